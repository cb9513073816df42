"""zero_g's `WnnCircuit` restated: the constraint system `WnnChip::configure` builds and the witness
`WnnChip::predict` lays out (/root/reference/src/gadgets/wnn.rs:125-237, 334-393), chip by chip with the
same names, column roles, gate / lookup creation order and region shapes, on halo2's SimpleFloorPlanner
(layouter.py).  The proving backend only ever sees its output -- the flat circuit image, the fixed and
sigma columns keygen would derive, and the six advice columns -- so this is the caller side of the hot
path, used to prove REAL inferences of the checked-in models in tests and bench.py.

Known deviations from a real halo2 keygen of the same circuit (neither changes satisfiability):
  * selector compression (`ConstraintSystem::compress_selectors`, run by keygen) is restated from its
    published algorithm in circuit.py; the merged columns it yields for this circuit cannot be compared
    with a real keygen here;
  * `LookupRangeCheckConfig` (halo2_gadgets v2023_04_20 utilities/lookup_range_check.rs) is not in the
    reference tree; it is restated from its published layout (K = 8, range_check.rs:10).
"""
from __future__ import annotations

import numpy as np

from circuit import ADVICE, FIXED, INSTANCE, R, ConstraintSystem
from layouter import AssignedCell, Layouter, Region
from symint import as_int_check, bit_of_byte, byte_be, eq, exact_shr, ge, gt, is_sym
from wnn_model import Wnn, WnnCircuitParams

K = 8  # bits per range-check word (range_check.rs:10)


def inv(x: int) -> int:
    return pow(x, -1, R)


# ------------------------------------------------------------------ bloom_filter/array_lookup.rs
def word_index_bits_for(n_hashes: int, bits_per_hash: int) -> int:
    """`impl From<BloomFilterConfig> for ArrayLookupConfig` (array_lookup.rs:50-75)."""
    assert bits_per_hash >= 7
    import math
    byte_index_bits = int((bits_per_hash - 3.0) / 2.0 - math.floor(math.log2(n_hashes)))
    return bits_per_hash - (byte_index_bits + 3)


class ArrayLookupChip:
    def __init__(self, config, bloom_filter_arrays: np.ndarray):
        self.config = config
        bph, wib = config["bits_per_hash"], config["word_index_bits"]
        assert bloom_filter_arrays.shape[1] == 1 << bph
        word_length = 1 << (bph - wib)
        weights = 1 << np.arange(word_length - 1, -1, -1, dtype=object)  # from_be_bits
        words = bloom_filter_arrays.reshape(bloom_filter_arrays.shape[0], -1, word_length).astype(object)
        self.bloom_filter_words = (words * weights).sum(axis=2)  # [bloom_index][word_index] -> int

    def word(self, bloom_index: int, word_index):
        """bloom_filter_words[bloom_index][word_index]; a recorded index becomes a table read of the tape"""
        if is_sym(word_index):
            tape = word_index.tape
            key = ("bloom_words", id(self))
            if key not in tape.__dict__.setdefault("table_bases", {}):
                tape.table_bases[key] = tape.add_table(self.bloom_filter_words.reshape(-1))
            base = tape.table_bases[key] + bloom_index * self.bloom_filter_words.shape[1]
            return tape.emit("TABLE", word_index.slot, imm=base, deps=(word_index.slot,))
        return int(self.bloom_filter_words[bloom_index][word_index])

    def bytes_per_word(self) -> int:
        return 1 << (self.config["bits_per_hash"] - self.config["word_index_bits"] - 3)

    @staticmethod
    def configure(meta: ConstraintSystem, hash_decomposition, byte_index, bit_index, bloom_index, bloom_value,
                  n_hashes: int, bits_per_hash: int, word_index_bits: int):
        assert bits_per_hash <= 32
        table_bloom_index, table_word_index, table_bloom_value = (meta.fixed_column() for _ in range(3))
        selector_id = meta.complex_selector()
        selector = meta.query_selector(selector_id)
        hd_cur = meta.advice(hash_decomposition, 0)
        hd_next = meta.advice(hash_decomposition, 1)
        byte_i = meta.advice(byte_index, 0)
        bit_i = meta.advice(bit_index, 0)
        current_hash = hd_cur - hd_next * (1 << bits_per_hash)
        word_index = (current_hash - byte_i * 8 - bit_i) * inv(1 << (bits_per_hash - word_index_bits))
        b_index = meta.advice(bloom_index, 0)
        b_value = meta.advice(bloom_value, 0)
        with_default = lambda x: selector * x + (1 - selector) * (R - 1)  # (-1, -1, -1) when inactive
        meta.lookup([with_default(b_index), with_default(word_index), with_default(b_value)],
                    [meta.fixed(table_bloom_index), meta.fixed(table_word_index), meta.fixed(table_bloom_value)])
        return dict(hash_decomposition=hash_decomposition, byte_index=byte_index, bit_index=bit_index,
                    bloom_index=bloom_index, bloom_value=bloom_value, selector=selector_id,
                    tables=(table_bloom_index, table_word_index, table_bloom_value),
                    n_hashes=n_hashes, bits_per_hash=bits_per_hash, word_index_bits=word_index_bits)

    def load(self, layouter: Layouter):
        rows = [(b, i, int(w)) for b, ws in enumerate(self.bloom_filter_words) for i, w in enumerate(ws)]
        rows.append((R - 1, R - 1, R - 1))
        layouter.assign_table(self.config["tables"], rows)

    def array_lookup(self, layouter: Layouter, hash_value: AssignedCell, bloom_index: int):
        c = self.config
        n_hashes, bph, wib = c["n_hashes"], c["bits_per_hash"], c["word_index_bits"]

        def region_fn(region: Region):
            h = hash_value.value
            hashes_le = [(h >> (i * bph)) & ((1 << bph) - 1) for i in range(n_hashes)]
            low = bph - wib
            idx = [(v >> low, (v & ((1 << low) - 1)) >> 3, v & 7) for v in hashes_le]
            words = [self.word(bloom_index, w) for w, _, _ in idx]
            decomposition = [h]
            for v in hashes_le:
                decomposition.append(exact_shr(decomposition[-1] - v, bph))  # (.. - v) * inv(2^bph)
            as_int_check(decomposition[-1], lambda x: x == 0, "hash does not fit n_hashes * bits_per_hash bits")
            for i, v in enumerate(decomposition):
                if i == 0:
                    hash_value.copy_advice(region, c["hash_decomposition"], 0)
                elif i < n_hashes:
                    region.assign_advice(c["hash_decomposition"], i, v)
                else:
                    region.assign_advice_from_constant(c["hash_decomposition"], i, 0)
            for i in range(n_hashes):
                region.assign_advice_from_constant(c["bloom_index"], i, bloom_index)
            word_cells = [region.assign_advice(c["bloom_value"], i, w) for i, w in enumerate(words)]
            byte_cells, bit_cells = [], []
            for i, (_, by, bi) in enumerate(idx):
                byte_cells.append(region.assign_advice(c["byte_index"], i, by))
                bit_cells.append(region.assign_advice(c["bit_index"], i, bi))
            for i in range(n_hashes):
                region.enable_selector(c["selector"], i)
            # big-endian order of the decomposition
            return [dict(word=w, byte_index=by, bit_index=bi)
                    for by, bi, w in reversed(list(zip(byte_cells, bit_cells, word_cells)))]

        return layouter.assign_region(region_fn)


# ------------------------------------------------------------------ bloom_filter/bit_selector.rs
class BitSelectorChip:
    def __init__(self, config):
        self.config = config

    @staticmethod
    def configure(meta: ConstraintSystem, byte, index, bit):
        selector_id = meta.complex_selector()
        byte_column, index_column, bit_column = (meta.fixed_column() for _ in range(3))
        s = meta.query_selector(selector_id)
        b, i, v = meta.advice(byte, 0), meta.advice(index, 0), meta.advice(bit, 0)
        # no default needed: (0, 0, 0) is in the table
        meta.lookup([s * b, s * i, s * v], [meta.fixed(byte_column), meta.fixed(index_column), meta.fixed(bit_column)])
        return dict(byte=byte, index=index, bit=bit, selector=selector_id, byte_column=byte_column,
                    index_column=index_column, bit_column=bit_column)

    def load(self, layouter: Layouter):
        rows = [(b, i, 0 if b & (1 << (7 - i)) == 0 else 1) for b in range(256) for i in range(8)]
        layouter.assign_table((self.config["byte_column"], self.config["index_column"], self.config["bit_column"]), rows)

    def select_bit(self, layouter: Layouter, byte: AssignedCell, index: AssignedCell) -> AssignedCell:
        c = self.config

        def region_fn(region: Region):
            bit = bit_of_byte(byte.value, index.value)  # (byte >> (7 - index)) & 1
            region.enable_selector(c["selector"], 0)
            byte.copy_advice(region, c["byte"], 0)
            index.copy_advice(region, c["index"], 0)
            return region.assign_advice(c["bit"], 0, bit)

        return layouter.assign_region(region_fn)


# ------------------------------------------------------------------ bloom_filter/byte_selector.rs
class ByteSelectorChip:
    def __init__(self, config):
        self.config = config

    @staticmethod
    def configure(meta: ConstraintSystem, byte_decomposition, lookup_index, byte_index, byte_selector, selector_acc,
                  byte_acc, byte_table):
        s_decomp = meta.complex_selector()
        s_is_bit, s_sel_acc, s_right, s_byte_acc = (meta.selector() for _ in range(4))

        def reconstruct_byte():
            z_cur = meta.advice(byte_decomposition, 0)
            z_next = meta.advice(byte_decomposition, 1)
            return z_cur - z_next * 256

        meta.lookup([meta.query_selector(s_decomp) * reconstruct_byte()], [meta.fixed(byte_table)])
        sel = meta.advice(byte_selector, 0)
        meta.create_gate([meta.query_selector(s_is_bit) * (sel * sel - sel)])                       # selector_is_bit
        q = meta.query_selector(s_sel_acc)
        meta.create_gate([q * (meta.advice(selector_acc, 1) - meta.advice(selector_acc, 0) - sel)])  # selector_acc
        q = meta.query_selector(s_right)
        meta.create_gate([q * (sel * (meta.advice(lookup_index, 0) - meta.advice(byte_index, 0)))])  # right_byte_selected
        q = meta.query_selector(s_byte_acc)
        acc_cur, acc_next = meta.advice(byte_acc, 0), meta.advice(byte_acc, 1)
        meta.create_gate([q * (acc_next - acc_cur - sel * reconstruct_byte())])                    # byte_acc
        return dict(byte_decomposition=byte_decomposition, lookup_index=lookup_index, byte_index=byte_index,
                    byte_selector=byte_selector, selector_acc=selector_acc, byte_acc=byte_acc,
                    selectors=(s_decomp, s_is_bit, s_sel_acc, s_right, s_byte_acc))

    def select_byte(self, layouter: Layouter, word: AssignedCell, index: AssignedCell, num_bytes: int) -> AssignedCell:
        c = self.config

        def region_fn(region: Region):
            w, idx = word.value, index.value
            bytes_be = [(w >> (8 * (num_bytes - 1 - i))) & 0xFF for i in range(num_bytes)]
            ith_byte = byte_be(w, idx, num_bytes)  # bytes_be[idx]
            decomposition = [w]
            for b in reversed(bytes_be):
                decomposition.append(exact_shr(decomposition[-1] - b, 8))  # (.. - b) * inv(256)
            as_int_check(decomposition[-1], lambda x: x == 0, "word does not fit num_bytes bytes")
            for i, v in enumerate(decomposition):
                if i == 0:
                    word.copy_advice(region, c["byte_decomposition"], 0)
                elif i < num_bytes:
                    region.assign_advice(c["byte_decomposition"], i, v)
                else:
                    region.assign_advice_from_constant(c["byte_decomposition"], i, 0)
            for i in range(num_bytes):
                index.copy_advice(region, c["lookup_index"], i)
            for i in range(num_bytes):  # little-endian rows, big-endian index
                region.assign_advice_from_constant(c["byte_index"], num_bytes - 1 - i, i)
            for i in range(num_bytes):
                region.assign_advice(c["byte_selector"], i, eq(idx, num_bytes - 1 - i))
            for i in range(num_bytes + 1):
                if i == 0:
                    region.assign_advice_from_constant(c["selector_acc"], 0, 0)
                elif i < num_bytes:
                    region.assign_advice(c["selector_acc"], i, ge(idx, num_bytes - i))
                else:
                    region.assign_advice_from_constant(c["selector_acc"], i, 1)
            result = region.assign_advice_from_constant(c["byte_acc"], 0, 0)
            for i in range(1, num_bytes + 1):
                result = region.assign_advice(c["byte_acc"], i, ge(idx, num_bytes - i) * ith_byte)
            for s in c["selectors"]:
                for i in range(num_bytes):
                    region.enable_selector(s, i)
            return result

        return layouter.assign_region(region_fn)


# ------------------------------------------------------------------ bloom_filter/and_bits.rs
class AndBitsChip:
    def __init__(self, config):
        self.config = config

    @staticmethod
    def configure(meta: ConstraintSystem, bits, acc):
        s = meta.selector()
        q = meta.query_selector(s)
        bit, acc_cur, acc_next = meta.advice(bits, 0), meta.advice(acc, 0), meta.advice(acc, 1)
        meta.create_gate([q * (acc_cur * bit - acc_next)])  # validate_bit_acc
        return dict(bits=bits, acc=acc, selector=s)

    def and_bits(self, layouter: Layouter, bits: list) -> AssignedCell:
        c = self.config

        def region_fn(region: Region):
            acc = [1]
            for b in bits:
                acc.append(acc[-1] * b.value % R)
            for i, b in enumerate(bits):
                b.copy_advice(region, c["bits"], i)
            cell = region.assign_advice_from_constant(c["acc"], 0, 1)
            for i in range(1, len(acc)):
                cell = region.assign_advice(c["acc"], i, acc[i])
                region.enable_selector(c["selector"], i - 1)
            return cell

        return layouter.assign_region(region_fn)


# ------------------------------------------------------------------ bloom_filter.rs
class BloomFilterChip:
    def __init__(self, config, bloom_filter_arrays: np.ndarray):
        self.array_lookup_chip = ArrayLookupChip(config["array_lookup"], bloom_filter_arrays)
        self.byte_selector_chip = ByteSelectorChip(config["byte_selector"])
        self.bit_selector_chip = BitSelectorChip(config["bit_selector"])
        self.and_bits_chip = AndBitsChip(config["and_bits"])

    @staticmethod
    def configure(meta: ConstraintSystem, a, n_hashes: int, bits_per_hash: int):
        array_lookup = ArrayLookupChip.configure(meta, a[0], a[1], a[2], a[3], a[4], n_hashes, bits_per_hash,
                                                 word_index_bits_for(n_hashes, bits_per_hash))
        bit_selector = BitSelectorChip.configure(meta, a[0], a[1], a[2])
        byte_column = bit_selector["byte_column"]  # shared with the byte selector and the range check
        byte_selector = ByteSelectorChip.configure(meta, a[0], a[1], a[2], a[3], a[4], a[5], byte_column)
        and_bits = AndBitsChip.configure(meta, a[4], a[5])
        return dict(array_lookup=array_lookup, byte_selector=byte_selector, bit_selector=bit_selector,
                    and_bits=and_bits, byte_column=byte_column)

    def load(self, layouter: Layouter):
        self.array_lookup_chip.load(layouter)
        self.bit_selector_chip.load(layouter)

    def bloom_lookup(self, layouter: Layouter, hash_value: AssignedCell, bloom_index: int) -> AssignedCell:
        bits = []
        for res in self.array_lookup_chip.array_lookup(layouter, hash_value, bloom_index):
            byte = self.byte_selector_chip.select_byte(layouter, res["word"], res["byte_index"],
                                                       self.array_lookup_chip.bytes_per_word())
            bits.append(self.bit_selector_chip.select_bit(layouter, byte, res["bit_index"]))
        return self.and_bits_chip.and_bits(layouter, bits)


# ------------------------------------------------------------------ range_check.rs (+ halo2_gadgets LookupRangeCheckConfig)
class RangeCheckConfig:
    def __init__(self, config):
        self.config = config

    @staticmethod
    def configure(meta: ConstraintSystem, advice_column, byte_column):
        # LookupRangeCheckConfig::configure(meta, running_sum = advice_column, table_idx = byte_column)
        meta.enable_equality(ADVICE, advice_column)
        q_lookup_id, q_running_id = meta.complex_selector(), meta.complex_selector()
        q_bitshift_id = meta.selector()
        q_lookup, q_running = meta.query_selector(q_lookup_id), meta.query_selector(q_running_id)
        z_cur = meta.advice(advice_column, 0)
        z_next = meta.advice(advice_column, 1)
        running_sum_lookup = q_running * (z_cur - z_next * (1 << K))
        short_lookup = (1 - q_running) * z_cur
        meta.lookup([q_lookup * (running_sum_lookup + short_lookup)], [meta.fixed(byte_column)])
        word = meta.advice(advice_column, -1)
        meta.create_gate([meta.query_selector(q_bitshift_id) * (word * (1 << K) * z_next - z_cur)])  # Short lookup bitshift
        # range_check.rs:34-51
        le_id = meta.selector()
        meta.create_gate([meta.query_selector(le_id) * (word + z_next - z_cur)])                     # le
        return dict(advice_column=advice_column, q_lookup=q_lookup_id, q_running=q_running_id,
                    q_bitshift=q_bitshift_id, le_selector=le_id)

    # LookupRangeCheckConfig::copy_check / range_check
    def copy_check(self, layouter: Layouter, element: AssignedCell, num_words: int, strict: bool) -> list:
        c = self.config

        def region_fn(region: Region):
            z = element.copy_advice(region, c["advice_column"], 0)
            zs = [z]
            for idx in range(num_words):
                word = z.value & ((1 << K) - 1)
                region.enable_selector(c["q_lookup"], idx)
                region.enable_selector(c["q_running"], idx)
                z = region.assign_advice(c["advice_column"], idx + 1, exact_shr(z.value - word, K))  # (z - word) * inv(2^K)
                zs.append(z)
            if strict:
                region.constrain_constant(zs[-1], 0)
            return zs

        return layouter.assign_region(region_fn)

    # LookupRangeCheckConfig::copy_short_check / short_range_check
    def copy_short_check(self, layouter: Layouter, element: AssignedCell, num_bits: int):
        c = self.config
        assert 0 < num_bits < K

        def region_fn(region: Region):
            e = element.copy_advice(region, c["advice_column"], 0)
            region.enable_selector(c["q_lookup"], 0)
            region.enable_selector(c["q_lookup"], 1)
            region.enable_selector(c["q_bitshift"], 1)
            region.assign_advice(c["advice_column"], 1, e.value * (1 << (K - num_bits)) % R)
            region.assign_advice_from_constant(c["advice_column"], 2, inv(1 << num_bits))

        layouter.assign_region(region_fn)

    def range_check(self, layouter: Layouter, input_cell: AssignedCell, n_bits: int):
        words = n_bits // K
        last_word = input_cell
        if words > 0:
            last_word = self.copy_check(layouter, input_cell, words, n_bits % K == 0)[-1]
        if n_bits % K != 0:
            self.copy_short_check(layouter, last_word, n_bits % K)

    def le_constant(self, layouter: Layouter, x: AssignedCell, y: int):
        c = self.config

        def region_fn(region: Region):
            x.copy_advice(region, c["advice_column"], 0)
            region.assign_advice_from_constant(c["advice_column"], 1, y)
            diff = region.assign_advice(c["advice_column"], 2, (y - x.value) % R)
            region.enable_selector(c["le_selector"], 1)
            return diff

        diff_cell = layouter.assign_region(region_fn)
        self.range_check(layouter, diff_cell, int(y).bit_length())


# ------------------------------------------------------------------ greater_than.rs
class GreaterThanChip:
    def __init__(self, config, range_check: RangeCheckConfig):
        self.config, self.range_check_config = config, range_check

    @staticmethod
    def configure(meta: ConstraintSystem, x, y, diff, is_gt):
        s = meta.selector()
        q = meta.query_selector(s)
        xv, yv, dv, gv = (meta.advice(col, 0) for col in (x, y, diff, is_gt))
        meta.create_gate([q * (xv + dv - gv * 256 - yv)])  # x + diff = 256 * is_gt + y
        return dict(x=x, y=y, diff=diff, is_gt=is_gt, selector=s)

    def _greater_than(self, region: Region, x_cell: AssignedCell, y: int):
        assert y <= 255, "y must be less than 256!"
        c = self.config
        is_gt = gt(x_cell.value, y)  # 1 if x > y else 0
        diff = (256 * is_gt + y - x_cell.value) % R
        region.enable_selector(c["selector"], 0)
        region.assign_advice_from_constant(c["y"], 0, y)
        diff_cell = region.assign_advice(c["diff"], 0, diff)
        gt_cell = region.assign_advice(c["is_gt"], 0, is_gt)
        return diff_cell, gt_cell

    def greater_than_witness(self, layouter: Layouter, x, y: int):
        def region_fn(region: Region):
            x_cell = region.assign_advice(self.config["x"], 0, x)
            diff_cell, gt_cell = self._greater_than(region, x_cell, y)
            return x_cell, diff_cell, gt_cell

        x_cell, diff_cell, gt_cell = layouter.assign_region(region_fn)
        rc = self.range_check_config
        rc.range_check(layouter, x_cell, 8)
        rc.range_check(layouter, gt_cell, 1)
        rc.range_check(layouter, diff_cell, 8)
        return x_cell, gt_cell

    def greater_than_copy(self, layouter: Layouter, x: AssignedCell, y: int) -> AssignedCell:
        def region_fn(region: Region):
            x_cell = x.copy_advice(region, self.config["x"], 0)
            return self._greater_than(region, x_cell, y)

        diff_cell, gt_cell = layouter.assign_region(region_fn)
        rc = self.range_check_config
        rc.range_check(layouter, gt_cell, 1)
        rc.range_check(layouter, diff_cell, 8)
        return gt_cell


# ------------------------------------------------------------------ encode_image.rs
class EncodeImageChip:
    def __init__(self, config, range_check: RangeCheckConfig, binarization_thresholds: np.ndarray):
        self.config = config
        self.greater_than_chip = GreaterThanChip(config["greater_than"], range_check)
        self.binarization_thresholds = binarization_thresholds

    @staticmethod
    def configure(meta: ConstraintSystem, x, y, diff, is_gt):
        return dict(advice_column=is_gt, greater_than=GreaterThanChip.configure(meta, x, y, diff, is_gt))

    def encode_image(self, layouter: Layouter, image: np.ndarray) -> list:
        thr = self.binarization_thresholds
        width, height = thr.shape[0], thr.shape[1]
        intensity_cells, bit_cells = {}, []
        for b in range(thr.shape[2]):
            for i in range(width):
                for j in range(height):
                    threshold = int(thr[i, j, b])
                    assert threshold <= 256
                    if threshold == 0:  # bit is one whatever the intensity
                        col = self.config["advice_column"]
                        bit_cell = layouter.assign_region(lambda region: region.assign_advice_from_constant(col, 0, 1))
                    else:
                        t = threshold - 1  # >= via >
                        first = intensity_cells.get((i, j))
                        if first is None:
                            px = image[i, j]  # (a recorded pixel when the image is symbolic: witness_tape.SymImage)
                            x_cell, bit_cell = self.greater_than_chip.greater_than_witness(
                                layouter, px if is_sym(px) else int(px), t)
                            intensity_cells[(i, j)] = x_cell
                        else:
                            bit_cell = self.greater_than_chip.greater_than_copy(layouter, first, t)
                    bit_cells.append(bit_cell)
        return bit_cells


# ------------------------------------------------------------------ hash.rs
class HashChip:
    def __init__(self, config, range_check: RangeCheckConfig):
        self.config, self.range_check_config = config, range_check
        assert config["n_bits"] * 3 <= 253, "Field too small to store x^3!"

    @staticmethod
    def configure(meta: ConstraintSystem, input_, quotient, remainder, msb, hash_, p: int, l: int, n_bits: int):
        s = meta.selector()
        q = meta.query_selector(s)
        iv, qv, rv, mv, hv = (meta.advice(col, 0) for col in (input_, quotient, remainder, msb, hash_))
        meta.create_gate([q * (iv * iv * iv - (qv * p + rv)), q * (rv - (mv * (1 << l) + hv))])  # hash
        return dict(selector=s, input=input_, quotient=quotient, remainder=remainder, msb=msb, hash=hash_,
                    p=p, l=l, n_bits=n_bits)

    def hash(self, layouter: Layouter, input_cell: AssignedCell) -> AssignedCell:
        c = self.config
        p, l, n_bits = c["p"], c["l"], c["n_bits"]

        def region_fn(region: Region):
            region.enable_selector(c["selector"], 0)
            x = input_cell.copy_advice(region, c["input"], 0).value
            cubed = x * x * x % R
            quotient = cubed // p
            remainder = cubed - quotient * p
            msb = remainder >> l
            h = remainder - (msb << l)
            return (region.assign_advice(c["quotient"], 0, quotient), region.assign_advice(c["remainder"], 0, remainder),
                    region.assign_advice(c["msb"], 0, msb), region.assign_advice(c["hash"], 0, h))

        quotient, remainder, msb, output = layouter.assign_region(region_fn)
        rc = self.range_check_config
        rc.range_check(layouter, quotient, n_bits * 3 - l)
        rc.range_check(layouter, msb, 1)
        rc.le_constant(layouter, remainder, p - 1)
        return output


# ------------------------------------------------------------------ response_accumulator.rs
class ResponseAccumulatorChip:
    def __init__(self, config):
        self.config = config

    @staticmethod
    def configure(meta: ConstraintSystem, advice_columns):
        s = meta.selector()
        q = meta.query_selector(s)
        x1, x2, x3, x4 = (meta.advice(advice_columns[i], 0) for i in range(4))
        prev_acc, acc = meta.advice(advice_columns[4], 0), meta.advice(advice_columns[4], 1)
        meta.create_gate([q * (x1 + x2 + x3 + x4 + prev_acc - acc)])  # accumulate_responses
        return dict(advice_columns=list(advice_columns), selector=s)

    def accumulate_responses(self, layouter: Layouter, responses: list) -> AssignedCell:
        c = self.config

        def region_fn(region: Region):
            acc_cell = region.assign_advice_from_constant(c["advice_columns"][4], 0, 0)
            acc = 0
            n_rows = (len(responses) + 3) // 4
            for row in range(n_rows):
                region.enable_selector(c["selector"], row)
                for i in range(4):
                    index = row * 4 + i
                    if index < len(responses):
                        responses[index].copy_advice(region, c["advice_columns"][i], row)
                        acc += responses[index].value
                    else:
                        region.assign_advice_from_constant(c["advice_columns"][i], row, 0)
                acc_cell = region.assign_advice(c["advice_columns"][4], row + 1, acc)
            return acc_cell

        return layouter.assign_region(region_fn)


# ------------------------------------------------------------------ bits2num.rs
class Bits2NumChip:
    def __init__(self, config):
        self.config = config

    @staticmethod
    def configure(meta: ConstraintSystem, input_, accumulator):
        s = meta.selector()
        bit_val = meta.advice(input_, 0)
        prev_acc, cur_acc = meta.advice(accumulator, 0), meta.advice(accumulator, 1)
        meta.create_gate([meta.query_selector(s) * (cur_acc - (prev_acc * 2 + bit_val))])  # next_num_constraint
        return dict(selector=s, input=input_, accumulator=accumulator)

    def convert_be(self, layouter: Layouter, bits: list) -> AssignedCell:
        c = self.config
        assert len(bits) <= 253

        def region_fn(region: Region):
            num = 0
            cell = region.assign_advice_from_constant(c["accumulator"], 0, 0)
            for i, bit in enumerate(bits):
                region.enable_selector(c["selector"], i)
                num = num * 2 + bit.value
                cell = region.assign_advice(c["accumulator"], i + 1, num)
                bit.copy_advice(region, c["input"], i)
            return cell

        return layouter.assign_region(region_fn)

    def convert_le(self, layouter: Layouter, bits: list) -> AssignedCell:
        return self.convert_be(layouter, list(reversed(bits)))


# ------------------------------------------------------------------ wnn.rs: WnnChip / WnnCircuit
class WnnChip:
    def __init__(self, config, bloom_filter_arrays: np.ndarray, binarization_thresholds: np.ndarray,
                 input_permutation: np.ndarray):
        n_classes, n_inputs, n_entries = bloom_filter_arrays.shape
        flat = bloom_filter_arrays.reshape(n_classes * n_inputs, n_entries)
        rc = RangeCheckConfig(config["range_check"])
        self.encode_image_chip = EncodeImageChip(config["encode_image"], rc, binarization_thresholds)
        self.bits2num_chip = Bits2NumChip(config["bits2num"])
        self.hash_chip = HashChip(config["hash"], rc)
        self.bloom_filter_chip = BloomFilterChip(config["bloom_filter"], flat)
        self.response_accumulator_chip = ResponseAccumulatorChip(config["response_accumulator"])
        self.input_permutation = input_permutation
        self.config = config
        self.n_classes, self.n_inputs = n_classes, n_inputs

    @staticmethod
    def configure(meta: ConstraintSystem, a, params: WnnCircuitParams):
        bloom_filter = BloomFilterChip.configure(meta, a, params.n_hashes, params.bits_per_hash)
        range_check = RangeCheckConfig.configure(meta, a[5], bloom_filter["byte_column"])
        encode_image = EncodeImageChip.configure(meta, a[0], a[1], a[2], a[3])
        hash_ = HashChip.configure(meta, a[0], a[1], a[2], a[3], a[4], params.p, params.l, params.bits_per_filter)
        response_accumulator = ResponseAccumulatorChip.configure(meta, a[0:5])
        bits2num = Bits2NumChip.configure(meta, a[3], a[4])
        return dict(bloom_filter=bloom_filter, range_check=range_check, encode_image=encode_image, hash=hash_,
                    response_accumulator=response_accumulator, bits2num=bits2num)

    def load(self, layouter: Layouter):
        self.bloom_filter_chip.load(layouter)

    def predict(self, layouter: Layouter, image: np.ndarray) -> list:
        bit_cells = self.encode_image_chip.encode_image(layouter, image)
        permuted = [bit_cells[int(i)] for i in self.input_permutation]
        n = self.config["hash"]["n_bits"]
        joint_inputs = [self.bits2num_chip.convert_le(layouter, permuted[c:c + n])
                        for c in range(0, len(permuted) - n + 1, n)]
        assert len(joint_inputs) == self.n_inputs
        hashes = [self.hash_chip.hash(layouter, x) for x in joint_inputs]
        responses = []
        for c in range(self.n_classes):
            responses.append([self.bloom_filter_chip.bloom_lookup(layouter, h, c * len(hashes) + i)
                              for i, h in enumerate(hashes)])
        return [self.response_accumulator_chip.accumulate_responses(layouter, r) for r in responses]


class WnnCircuit:
    """`WnnCircuit::configure_with_params` + `synthesize` (wnn.rs:334-393)."""

    def __init__(self, wnn: Wnn, k: int, compress_selectors: bool = True):
        self.wnn, self.k, self.merge = wnn, k, compress_selectors
        self.params = wnn.get_circuit_params()
        cs = ConstraintSystem(k)
        self.instance_column = cs.instance_column()
        self.advice_columns = [cs.advice_column() for _ in range(6)]
        for a in self.advice_columns:
            cs.enable_equality(ADVICE, a)
        cs.enable_equality(INSTANCE, self.instance_column)
        self.constants = cs.fixed_column()
        cs.enable_equality(FIXED, self.constants)  # enable_constant
        self.config = WnnChip.configure(cs, self.advice_columns, self.params)
        self.cs = cs

    def synthesize(self, image: np.ndarray):
        """-> (Assignment, class scores).  The fixed / sigma columns do not depend on the image values
        (keygen runs the same synthesis on a zero image, wnn.rs:222-229) -- tests check that."""
        layouter = Layouter(self.cs, self.constants)
        chip = WnnChip(self.config, self.wnn.bloom_filters, self.wnn.binarization_thresholds,
                       self.wnn.input_permutation)
        chip.load(layouter)
        result = chip.predict(layouter, image)
        for i, score in enumerate(result):
            layouter.constrain_instance(score, self.instance_column, i, score.value)
        self.rows_used, self.n_regions = layouter.rows_used(), layouter.n_regions
        layouter.asg.compress_selectors(self.merge)  # keygen_vk: cs.compress_selectors(assembly.selectors)
        return layouter.asg, [s.value for s in result]


def build(wnn: Wnn, image: np.ndarray, k: int, compress_selectors: bool = True):
    """-> (ConstraintSystem, Assignment, instance length, scores): what tests / bench.py feed the prover."""
    circuit = WnnCircuit(wnn, k, compress_selectors)
    asg, scores = circuit.synthesize(image)
    return circuit.cs, asg, len(scores), scores
