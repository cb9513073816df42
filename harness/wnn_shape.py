"""A circuit of the SHAPE of zero_g's WnnCircuit, with a satisfying synthetic witness.

Column / gate / lookup inventory follows SURVEY.md appendix A, i.e. `WnnChip::configure`
(/root/reference/src/gadgets/wnn.rs:125-172) and the chips it wires:
  byte_selector  /root/reference/src/gadgets/bloom_filter/byte_selector.rs:105-165  (gates 1-4, lookup L2)
  and_bits       /root/reference/src/gadgets/bloom_filter/and_bits.rs:59-72         (gate 5)
  range check    /root/reference/src/gadgets/range_check.rs:31-51                   (gates 6-7, lookup L3)
  greater_than   /root/reference/src/gadgets/greater_than.rs:82-93                  (gate 8)
  hash           /root/reference/src/gadgets/hash.rs:84-107                         (gates 9-10)
  response_acc.  /root/reference/src/gadgets/response_accumulator.rs:55-67          (gate 11)
  bits2num       /root/reference/src/gadgets/bits2num.rs:55-66                      (gate 12)
  array_lookup   /root/reference/src/gadgets/bloom_filter/array_lookup.rs:184-233   (lookup L0)
  bit_selector   /root/reference/src/gadgets/bloom_filter/bit_selector.rs:108-123   (lookup L1)
6 advice + 1 instance + 23 fixed columns (constants, 6 table columns, 16 selectors kept uncompressed),
cs.degree() = 6, 8 equality columns -> 2 permutation sets, blinding_factors = 5.

The WITNESS is not the WNN inference (the Rust WnnChip stays the owner of that, SURVEY.md section 2
rows 12-13): rows are filled gadget by gadget with seeded random values that satisfy every gate and
lookup, which reproduces the column sparsity (bits, bytes, zeros) that matters to the MSM.
"""
from __future__ import annotations

import random

from circuit import ADVICE, FIXED, INSTANCE, R, Assignment, ConstraintSystem

# model_28input_256entry_1hash_1bpi: p = 509, l = 8, bph = 8, word_index_bits = 3
MODELS = {
    "tiny": dict(k=14, p=509, l=8, bph=8, wib=3, filters=280),
    "small": dict(k=15, p=2097143, l=20, bph=10, wib=5, filters=560),
    "medium": dict(k=15, p=8388593, l=22, bph=11, wib=5, filters=840),
    "large": dict(k=17, p=(1 << 52) - 47, l=52, bph=13, wib=7, filters=960),
}


def build(model: str = "tiny", k: int | None = None, seed: int = 0, copy_fraction: float = 0.3):
    m = dict(MODELS[model])
    if k is not None:
        m["k"] = k
    k = m["k"]
    rng = random.Random(seed)
    cs = ConstraintSystem(k)
    inst = cs.instance_column()
    a = [cs.advice_column() for _ in range(6)]
    const = cs.fixed_column()
    t_bidx, t_bword, t_bval = cs.fixed_column(), cs.fixed_column(), cs.fixed_column()
    t_byte, t_bitidx, t_bit = cs.fixed_column(), cs.fixed_column(), cs.fixed_column()
    # complex selectors
    q_array, q_bitsel, q_bytesel, q_lookup, q_running = (cs.fixed_column() for _ in range(5))
    # simple selectors (left uncompressed: one fixed column each)
    s1, s2, s3, s4, s5, s6, s7, s8, s9, s11, s12 = (cs.fixed_column() for _ in range(11))
    for c in a:
        cs.enable_equality(ADVICE, c)
    cs.enable_equality(INSTANCE, inst)
    cs.enable_equality(FIXED, const)

    A = lambda i, r=0: cs.advice(a[i], r)
    Fx = cs.fixed
    two8 = 256
    # ---- the 12 gate polynomials, creation order of appendix A
    cs.create_gate([Fx(s1) * (A(3) * A(3) - A(3))])
    cs.create_gate([Fx(s2) * (A(4, 1) - A(4) - A(3))])
    cs.create_gate([Fx(s3) * A(3) * (A(1) - A(2))])
    cs.create_gate([Fx(s4) * (A(5, 1) - A(5) - A(3) * (A(0) - two8 * A(0, 1)))])
    cs.create_gate([Fx(s5) * (A(5) * A(4) - A(5, 1))])
    cs.create_gate([Fx(s6) * (A(5, -1) * two8 * A(5, 1) - A(5))])
    cs.create_gate([Fx(s7) * (A(5, -1) + A(5, 1) - A(5))])
    cs.create_gate([Fx(s8) * (A(0) + A(2) - two8 * A(3) - A(1))])
    cs.create_gate([Fx(s9) * (A(0) * A(0) * A(0) - (m["p"] * A(1) + A(2))),
                    Fx(s9) * (A(2) - ((1 << m["l"]) * A(3) + A(4)))])
    cs.create_gate([Fx(s11) * (A(0) + A(1) + A(2) + A(3) + A(4) - A(4, 1))])
    cs.create_gate([Fx(s12) * (A(4, 1) - (2 * A(4) + A(3)))])
    # ---- the 4 lookups
    bph, wib = m["bph"], m["wib"]
    low = 1 << (bph - wib)
    inv_low = pow(low, -1, R)
    qa = Fx(q_array)
    word_index = ((A(0) - (1 << bph) * A(0, 1)) - 8 * A(1) - A(2)) * inv_low
    default = lambda e: qa * e + (1 - qa) * (R - 1)
    cs.lookup([default(A(3)), default(word_index), default(A(4))], [Fx(t_bidx), Fx(t_bword), Fx(t_bval)])
    qb = Fx(q_bitsel)
    cs.lookup([qb * A(0), qb * A(1), qb * A(2)], [Fx(t_byte), Fx(t_bitidx), Fx(t_bit)])
    cs.lookup([Fx(q_bytesel) * (A(0) - two8 * A(0, 1))], [Fx(t_byte)])
    ql, qr = Fx(q_lookup), Fx(q_running)
    cs.lookup([ql * (qr * (A(5) - two8 * A(5, 1)) + (1 - qr) * A(5))], [Fx(t_byte)])
    assert cs.degree() == 6 and cs.blinding_factors() == 5 and len(cs.perm_columns) == 8

    asg = Assignment(cs)
    n, usable = 1 << k, cs.usable_rows()
    # ---- tables: bloom (index, word_index, word) + default tuple; bit table byte x index (2048 rows)
    word_bits = 1 << (bph - wib)  # bits addressed inside one bloom word
    bloom = {}
    row = 0
    n_filters = min(m["filters"], (usable - 2) >> wib)
    for f in range(n_filters):
        for w in range(1 << wib):
            val = rng.getrandbits(min(word_bits, 32))
            bloom[(f, w)] = val
            asg.set(FIXED, t_bidx, row, f)
            asg.set(FIXED, t_bword, row, w)
            asg.set(FIXED, t_bval, row, val)
            row += 1
    for c in (t_bidx, t_bword, t_bval):  # default tuple, then padding rows repeat it
        for r in range(row, usable):
            asg.set(FIXED, c, r, R - 1)
    for r in range(usable):
        b, i = (r >> 3) & 0xFF, r & 7
        if r >= 2048:
            b, i = 0, 0  # padding rows repeat the (0, 0, 0) entry
        asg.set(FIXED, t_byte, r, b)
        asg.set(FIXED, t_bitidx, r, i)
        asg.set(FIXED, t_bit, r, (b >> i) & 1)
    assert usable >= 2048 + 8, "k too small for the 2048-row bit table"

    S = lambda col, r, v: asg.set(ADVICE, a[col], r, v)
    on = lambda sel, r: asg.set(FIXED, sel, r, 1)

    # ---- gadget regions; each returns the number of rows it used
    def byte_selector(r):
        nbytes = 4
        word = rng.getrandbits(8 * nbytes)
        pick = rng.randrange(nbytes)
        cnt = acc = 0
        for j in range(nbytes):
            bit = 1 if j == pick else 0
            S(0, r + j, word >> (8 * j)); S(1, r + j, j); S(2, r + j, pick); S(3, r + j, bit)
            S(4, r + j, cnt); S(5, r + j, acc)
            for sel in (s1, s2, s3, s4, q_bytesel):
                on(sel, r + j)
            cnt += bit
            acc += bit * ((word >> (8 * j)) & 0xFF)
        S(0, r + nbytes, 0); S(4, r + nbytes, cnt); S(5, r + nbytes, acc)
        return nbytes + 1

    def and_bits(r):
        m_ = 3
        acc = 1
        for j in range(m_):
            bit = rng.getrandbits(1)
            S(4, r + j, bit); S(5, r + j, acc)
            on(s5, r + j)
            acc *= bit
        S(5, r + m_, acc)
        return m_ + 1

    def bit_selector(r):
        b, i = rng.getrandbits(8), rng.randrange(8)
        S(0, r, b); S(1, r, i); S(2, r, (b >> i) & 1)
        on(q_bitsel, r)
        return 1

    def array_lookup(r):
        f, w = rng.randrange(n_filters), rng.randrange(1 << wib)
        byte_i, bit_i = rng.randrange(low // 8 if low >= 8 else 1), rng.randrange(8 if low >= 8 else low)
        hsh = w * low + 8 * byte_i + bit_i
        rest = rng.getrandbits(8)
        S(0, r, hsh + (rest << bph)); S(0, r + 1, rest)
        S(1, r, byte_i); S(2, r, bit_i); S(3, r, f); S(4, r, bloom[(f, w)])
        on(q_array, r)
        return 2

    def range_check_running(r):
        words = 3
        z = rng.getrandbits(8 * words)
        for j in range(words):
            S(5, r + j, z)
            on(q_lookup, r + j); on(q_running, r + j)
            z >>= 8
        S(5, r + words, 0)
        return words + 1

    def range_check_short(r):
        s_bits = rng.randrange(1, 8)
        word = rng.getrandbits(s_bits)
        S(5, r, word); on(q_lookup, r)
        S(5, r + 1, word << (8 - s_bits)); on(q_lookup, r + 1); on(s6, r + 1)
        S(5, r + 2, pow(1 << s_bits, -1, R))
        return 3

    def less_equal(r):
        x, d = rng.getrandbits(8), rng.getrandbits(8)
        S(5, r, x); S(5, r + 1, x + d); S(5, r + 2, d)
        on(s7, r + 1)
        return 3

    def greater_than(r):
        x, y = rng.getrandbits(8), rng.getrandbits(8)
        gt = 1 if x > y else 0
        S(0, r, x); S(1, r, y); S(3, r, gt); S(2, r, 256 * gt + y - x)
        on(s8, r)
        return 1

    def hash_row(r):
        x = rng.getrandbits(28)
        q_, rem = divmod(x ** 3, m["p"])
        S(0, r, x); S(1, r, q_); S(2, r, rem); S(3, r, rem >> m["l"]); S(4, r, rem & ((1 << m["l"]) - 1))
        on(s9, r)
        return 1

    def accumulate(r):
        v = [rng.getrandbits(1) for _ in range(5)]
        for c in range(5):
            S(c, r, v[c])
        S(4, r + 1, sum(v))
        on(s11, r)
        return 2

    def bits2num(r):
        m_ = 4
        acc = 0
        for j in range(m_):
            bit = rng.getrandbits(1)
            S(3, r + j, bit); S(4, r + j, acc)
            on(s12, r + j)
            acc = 2 * acc + bit
        S(4, r + m_, acc)
        return m_ + 1

    # mix weighted like the real layout: bloom-filter gadgets dominate, then thresholds / range checks
    gadgets = ([byte_selector] * 6 + [array_lookup] * 4 + [bit_selector] * 4 + [and_bits] * 3 +
               [range_check_running] * 4 + [range_check_short] * 2 + [less_equal] * 2 + [greater_than] * 4 +
               [hash_row] * 1 + [accumulate] * 1 + [bits2num] * 1)
    r = 1  # row 0 stays empty so that rotation -1 of the first region reads a zero row
    fill = int(usable * 0.93)
    while r < fill - 8:
        r += rng.choice(gadgets)(r) + 1  # one spacer row between regions
    # ---- public outputs + constants + copy constraints between equal cells
    scores = [rng.randrange(30) for _ in range(10)]
    for i, sc in enumerate(scores):
        asg.set(INSTANCE, inst, i, sc)
        S(4, fill + i, sc)
        asg.copy((ADVICE, a[4], fill + i), (INSTANCE, inst, i))
    asg.set(FIXED, const, 0, 0)
    asg.set(FIXED, const, 1, 1)
    by_value = {}
    for c in range(6):
        col = asg.advice[c]
        for rr in range(1, fill):
            by_value.setdefault(col[rr], []).append((ADVICE, a[c], rr))
    for val, cells in by_value.items():
        rng.shuffle(cells)
        take = int(len(cells) * copy_fraction)
        for x, y in zip(cells[:take], cells[1:take + 1]):
            asg.copy(x, y)
        if val in (0, 1) and cells:
            asg.copy(cells[0], (FIXED, const, val))
    return cs, asg, 10


if __name__ == "__main__":
    import sys
    import time

    kk = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    t = time.time()
    cs_, asg_, _ = build("tiny", k=kk)
    print("built k=%d in %.1fs: %d fixed, %d queries, degree %d" % (kk, time.time() - t, cs_.n_fixed, len(cs_.queries), cs_.degree()))
    t = time.time()
    asg_.check()
    print("witness satisfies all gates and lookups (%.1fs)" % (time.time() - t))
