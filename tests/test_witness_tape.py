"""The recorded witness program (harness/witness_tape.py): WnnChip::predict traced once on a symbolic image must give,
replayed on concrete images, exactly the advice columns and class scores the concrete synthesis gives -- the same chip
code produces both.  The flat form handed to zg_witness_plan_create is checked for the properties the device relies
on.  (The device's replay against this interpreter: tests/test_gpu_witness.py.)"""
import numpy as np
import pytest

import symint
import witness_tape
import wnn_circuit
import wnn_model


def _images(n, seed=5):
    real = wnn_model.load_test_image()
    rng = np.random.default_rng(seed)
    extremes = [np.zeros_like(real), np.full_like(real, 255)]
    return [real] + extremes + [rng.integers(0, 256, size=real.shape, dtype=real.dtype) for _ in range(n)]


@pytest.mark.parametrize("which", [wnn_model.MNIST_TINY, wnn_model.MNIST_SMALL])
def test_replay_equals_concrete_synthesis(which):
    k, name = which
    wnn = wnn_model.load_checked_in(name)
    prog = witness_tape.trace(wnn, k)
    assert prog.image_bytes == 28 * 28 and prog.n_advice == 6
    for im in _images(2 if k == 14 else 1):
        cs, asg, ilen, scores = wnn_circuit.build(wnn, im, k)
        adv, got_scores = prog.run(im)
        assert got_scores == scores == wnn.predict(im)
        assert adv == [[int(v) for v in col] for col in asg.advice]


def test_replay_of_the_k17_stand_in():
    """49-bit filter inputs: cubes of 147 bits, a 53-bit hash modulus"""
    wnn = wnn_model.synthetic_wnn()
    prog = witness_tape.trace(wnn, 17)
    im = wnn_model.load_test_image()
    cs, asg, ilen, scores = wnn_circuit.build(wnn, im, 17)
    adv, got_scores = prog.run(im)
    assert got_scores == scores == wnn.predict(im)
    assert adv == [[int(v) for v in col] for col in asg.advice]


def test_flat_form_is_straight_line():
    k, name = wnn_model.MNIST_TINY
    prog = witness_tape.trace(wnn_model.load_checked_in(name), k)
    a = prog.arrays()
    ops, ls = a["ops"], a["level_start"]
    assert ls[0] == 0 and ls[-1] == ops.shape[0] and np.all(np.diff(ls.astype(np.int64)) >= 0)
    level_of = np.repeat(np.arange(len(ls) - 1), np.diff(ls.astype(np.int64)))
    first = ls[level_of]  # first op of the op's own level: operands must lie before it
    code = ops[:, 0]
    uses_a = code >= symint.OPCODE["ADD"]
    uses_b = np.isin(code, [symint.OPCODE[x] for x in ("ADD", "SUB", "MUL", "SHRV")])
    assert np.all(ops[uses_a, 1] < first[uses_a]) and np.all(ops[uses_b, 2] < first[uses_b])
    assert np.all(ops[code == symint.OPCODE["PIXEL"], 3] < a["image_bytes"])
    assert np.all(ops[code == symint.OPCODE["CONST"], 3] < a["consts"].shape[0])
    assert np.all(ops[code == symint.OPCODE["TABLE"], 3] < a["table"].shape[0])
    cs = a["cell_slot"]
    assert cs.shape == (6, 1 << k)
    assigned = cs != witness_tape.NO_SLOT
    assert np.all(cs[assigned] < ops.shape[0]) and assigned.sum() == len(prog.cells)
    assert a["instance_slots"].shape == (10,)
    # renumbering is consistent: replaying the flat form gives the recorded program's values
    t = symint.Tape()
    t.ops = [(int(o), int(x), int(y), int(i)) for o, x, y, i in ops]
    t.consts = [sum(int(w) << (64 * j) for j, w in enumerate(c)) for c in a["consts"]]
    t.table = [int(w) for w in a["table"]]
    im = wnn_model.load_test_image()
    v = t.run(im.reshape(-1))
    assert [v[s] for s in a["instance_slots"]] == [9, 6, 13, 10, 17, 10, 9, 26, 11, 16]  # integration_test.rs:19


def test_helpers_on_integers_are_the_reference_formulas():
    """On Python integers the chip helpers must compute what the reference's closures compute -- the running-sum step as a
    product with the field inverse, selections as comparisons -- and on recorded values the same numbers."""
    import random

    rnd = random.Random(9)
    inv = lambda x: pow(x, -1, symint.R)
    t = symint.Tape()
    for _ in range(50):
        k = rnd.choice((1, 3, 8, 13))
        x = rnd.getrandbits(60) << k
        assert symint.exact_shr(x, k) == x * inv(1 << k) % symint.R == x >> k
        w, nb = rnd.getrandbits(64), 8
        idx = rnd.randrange(nb)
        assert symint.byte_be(w, idx, nb) == [(w >> (8 * (nb - 1 - i))) & 0xFF for i in range(nb)][idx]
        byte, bit = rnd.getrandbits(8), rnd.randrange(8)
        assert symint.bit_of_byte(byte, bit) == (byte >> (7 - bit)) & 1
        y = rnd.getrandbits(8)
        assert (symint.gt(byte, y), symint.ge(byte, y), symint.eq(byte, y)) == (int(byte > y), int(byte >= y), int(byte == y))
        # ... and recorded: the same values out of the interpreter
        sw, si, sb, sbit = t.const(w), t.const(idx), t.const(byte), t.const(bit)
        rec = [symint.exact_shr(t.const(x), k), symint.byte_be(sw, si, nb), symint.byte_be(sw, idx, nb), symint.bit_of_byte(sb, sbit),
               symint.bit_of_byte(sb, bit), symint.gt(sb, y), symint.ge(sb, y), symint.eq(sb, y), 256 * symint.gt(sb, y) + y - sb + 255]
        v = t.run([])
        assert [v[r.slot] for r in rec] == [x >> k, symint.byte_be(w, idx, nb), symint.byte_be(w, idx, nb), (byte >> (7 - bit)) & 1,
                                             (byte >> (7 - bit)) & 1, int(byte > y), int(byte >= y), int(byte == y),
                                             256 * int(byte > y) + y - byte + 255]


@pytest.mark.parametrize("which", [wnn_model.MNIST_TINY, wnn_model.MNIST_MEDIUM])
def test_parallel_prefix_form_shows_the_same_values_in_a_third_of_the_levels(which):
    """harness/tape_opt.py (round 5): the Horner and running-sum chains of the gadgets as Kogge-Stone prefixes.  Every cell and
    every class score of the optimised program equals the recorded program's for every image; the level count -- what the
    device's replay time is proportional to -- falls to a third."""
    k, name = which
    wnn = wnn_model.load_checked_in(name)
    plain = witness_tape.trace(wnn, k, optimised=False)
    fast = witness_tape.optimise(plain)
    assert fast.opt["levels_before"] == max(plain.tape.level) + 1 and fast.opt["levels_after"] == max(fast.tape.level) + 1
    assert fast.opt["levels_after"] * 5 <= fast.opt["levels_before"] * 2 and fast.opt["chains"] >= 30
    assert fast.opt["operations_after"] < 1.1 * fast.opt["operations_before"]
    assert set(fast.cells) == set(plain.cells) and set(fast.instances) == set(plain.instances)
    for im in _images(2 if k == 14 else 0):
        assert fast.run(im) == plain.run(im)
    assert witness_tape.trace(wnn, k).opt == fast.opt  # (trace() optimises by default)


def test_parallel_prefix_on_synthetic_chains():
    """mixed links (ADD, ADDI, through MULI or not), a chain whose multipliers outgrow 64 bits (left alone), a link whose
    predecessor has two successors (the chain forks: one branch continues it, the other starts anew), wrap-around modulo 2^256"""
    import random

    import tape_opt

    rnd = random.Random(9)
    t = symint.Tape()
    px = [t.pixel(i) for i in range(8)]
    outs = []
    x = px[0]
    for j in range(40):  # x = 3 x + p (ADD) / 3 x + 7 (ADDI) / x + p
        kind = j % 3
        x = x * 3 + px[j % 8] if kind == 0 else (x * 3 + 7 if kind == 1 else x + px[(j * 5) % 8])
        outs.append(x)
    y = px[1]
    for j in range(12):  # multipliers 2^40 each: 2^80 after one round -- must be left as recorded
        y = y * (1 << 40) + px[j % 8]
        outs.append(y)
    z = px[2]
    fork = None
    for j in range(10):
        z = z + px[j % 8]
        outs.append(z)
        if j == 4:
            fork = z
    w = fork
    for j in range(8):
        w = w * 2 + px[j]
        outs.append(w)
    big = t.emit("CONST", imm=0)
    t.consts.append((1 << 256) - 5)
    v = symint.Sym(t, big.slot)
    for j in range(6):  # sums that wrap modulo 2^256
        v = v + px[j]
        outs.append(v)
    prog = witness_tape.WitnessProgram(t, {(0, i): o.slot for i, o in enumerate(outs)}, {0: outs[-1].slot}, 1, 7, 8)
    fast = witness_tape.optimise(prog)
    assert fast.opt["chains"] >= 3 and fast.opt["levels_after"] < fast.opt["levels_before"]
    for _ in range(5):
        im = np.array([rnd.randrange(256) for _ in range(8)], dtype=np.uint8)
        a = tape_opt_run(prog, im)
        b = tape_opt_run(fast, im)
        assert a == b


def tape_opt_run(prog, image):
    """values of every shown slot, computed modulo 2^256 without the tape contract's no-underflow assertions"""
    v = [0] * len(prog.tape.ops)
    t = prog.tape
    for i, (op, a, b, imm) in enumerate(t.ops):
        n = symint.OPS[op]
        r = {"CONST": lambda: t.consts[imm], "PIXEL": lambda: int(image[imm]), "ADD": lambda: v[a] + v[b], "MULI": lambda: v[a] * imm,
             "ADDI": lambda: v[a] + imm}[n]()
        v[i] = r & symint.M256
    return {c: v[s] for c, s in prog.cells.items()}
