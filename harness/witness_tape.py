"""The witness of zero_g's WnnCircuit as a straight-line program over the image bytes.

`trace(wnn, k)` runs `WnnChip::predict` (wnn_circuit.py, the same chip code that synthesises a concrete witness) once
on a symbolic image: every advice cell ends up as a tape slot (symint.Tape), the floor planner places the regions as
it does for a concrete image (the layout does not depend on the pixel values), and the class scores are the slots
constrained to the instance column.  `WitnessProgram.arrays()` is the flat form the device takes
(include/zg_halo2.h zg_witness_plan_create): operations sorted by dependency level, the constant pool, the bloom
filter words, and for every advice cell the slot it shows (or none: the cell stays zero).
"""
from __future__ import annotations

import numpy as np

from layouter import AssignedCell, Layouter, Region
from symint import OPCODE, R, Sym, Tape, is_sym
from wnn_circuit import WnnChip, WnnCircuit
from wnn_model import Wnn

NO_SLOT = 0xFFFFFFFF


class SymImage:
    """image[i, j] -> the recorded byte at offset i * width + j of a row-major 8-bit image"""

    def __init__(self, tape: Tape, height: int, width: int):
        self.tape, self.shape = tape, (height, width)
        self._px = {}

    def __getitem__(self, ij):
        i, j = ij
        if ij not in self._px:
            self._px[ij] = self.tape.pixel(i * self.shape[1] + j)
        return self._px[ij]


class TapeLayouter(Layouter):
    """Places regions exactly as Layouter does, but records (column, row, slot) instead of writing values."""

    def __init__(self, cs, constants_column: int, tape: Tape):
        super().__init__(cs, constants_column)
        self.tape = tape
        self.cells = {}       # (column, row) -> slot
        self.instances = {}   # instance row -> slot

    def slot_of(self, value) -> int:
        return value.slot if is_sym(value) else self.tape.const(int(value)).slot

    def commit_region(self, region: Region):
        for c in region.cells:
            self.cells[(c.column, c.row)] = self.slot_of(c.value)
        key = ("f", self.constants_column)  # (the constants column advances as in a concrete synthesis)
        self.columns[key] = self.columns.get(key, 0) + len(region.constants)

    def assign_table(self, columns, rows):
        pass  # fixed columns: proving-key material, not witness

    def constrain_instance(self, cell: AssignedCell, instance_column: int, row: int, value):
        self.instances[row] = self.slot_of(cell.value)


class WitnessProgram:
    def __init__(self, tape: Tape, cells: dict, instances: dict, n_advice: int, k: int, image_bytes: int):
        self.tape, self.cells, self.instances = tape, cells, instances
        self.n_advice, self.k, self.image_bytes = n_advice, k, image_bytes

    def run(self, image: np.ndarray):
        """reference interpreter -> (advice[n_advice][2^k] as Python integers, class scores)"""
        v = self.tape.run(np.asarray(image, dtype=np.uint8).reshape(-1))
        n = 1 << self.k
        adv = [[0] * n for _ in range(self.n_advice)]
        for (col, row), slot in self.cells.items():
            adv[col][row] = v[slot] % R
        return adv, [v[self.instances[i]] for i in range(len(self.instances))]

    def arrays(self) -> dict:
        """flat arrays for zg_witness_plan_create: ops in level order (slots renumbered accordingly)"""
        t = self.tape
        # (within a level by opcode: the lanes of a wave then mostly run the same case of the interpreter's switch)
        order = sorted(range(len(t.ops)), key=lambda i: (t.level[i], t.ops[i][0], i))
        new = {old: i for i, old in enumerate(order)}
        ops = np.zeros((len(order), 4), dtype=np.uint64)  # opcode, a, b, imm
        uses_a = {OPCODE[x] for x in ("ADD", "SUB", "MUL", "ADDI", "RSUBI", "MULI", "SHRI", "SHLI", "ANDI", "SHRV", "GTI",
                                      "GEI", "EQI", "DIVI", "TABLE")}
        uses_b = {OPCODE[x] for x in ("ADD", "SUB", "MUL", "SHRV")}
        for i, old in enumerate(order):
            op, a, b, imm = t.ops[old]
            ops[i] = (op, new[a] if op in uses_a else 0, new[b] if op in uses_b else 0, imm)
        levels = [t.level[old] for old in order]
        n_levels = (max(levels) + 1) if levels else 0
        level_start = np.searchsorted(np.array(levels), np.arange(n_levels + 1)).astype(np.uint32)
        consts = np.zeros((max(1, len(t.consts)), 4), dtype=np.uint64)
        for i, c in enumerate(t.consts):
            for w in range(4):
                consts[i, w] = (c >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
        n = 1 << self.k
        cell_slot = np.full((self.n_advice, n), NO_SLOT, dtype=np.uint32)
        for (col, row), slot in self.cells.items():
            cell_slot[col, row] = new[slot]
        inst = np.array([new[self.instances[i]] for i in range(len(self.instances))], dtype=np.uint32)
        return dict(ops=ops, level_start=level_start, consts=consts,
                    table=np.array(t.table if t.table else [0], dtype=np.uint64), cell_slot=cell_slot,
                    instance_slots=inst, image_bytes=self.image_bytes)


def optimise(prog: "WitnessProgram") -> "WitnessProgram":
    """The same program with its serial chains as parallel prefixes (tape_opt.py): every cell and instance row shows the same
    value for every image, in a third of the levels.  `prog.opt` says what was done."""
    import tape_opt

    ops, level, stats = tape_opt.parallelise_chains(prog.tape)
    roots = set(prog.cells.values()) | set(prog.instances.values())
    tape, new = tape_opt.rebuild(prog.tape, ops, level, roots)
    out = WitnessProgram(tape, {c: new[s] for c, s in prog.cells.items()}, {i: new[s] for i, s in prog.instances.items()},
                         prog.n_advice, prog.k, prog.image_bytes)
    stats.update(levels_after=max(tape.level) + 1, operations_before=len(prog.tape.ops), operations_after=len(tape.ops))
    out.opt = stats
    return out


def trace(wnn: Wnn, k: int, compress_selectors: bool = True, optimised: bool = True) -> WitnessProgram:
    circuit = WnnCircuit(wnn, k, compress_selectors)
    tape = Tape()
    layouter = TapeLayouter(circuit.cs, circuit.constants, tape)
    chip = WnnChip(circuit.config, wnn.bloom_filters, wnn.binarization_thresholds, wnn.input_permutation)
    chip.load(layouter)
    height, width = wnn.binarization_thresholds.shape[0], wnn.binarization_thresholds.shape[1]
    result = chip.predict(layouter, SymImage(tape, height, width))
    for i, score in enumerate(result):
        layouter.constrain_instance(score, circuit.instance_column, i, score.value)
    prog = WitnessProgram(tape, layouter.cells, layouter.instances, len(circuit.advice_columns), k, height * width)
    return optimise(prog) if optimised else prog
