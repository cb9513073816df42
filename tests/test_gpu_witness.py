"""The witness of a batch of images on the device (zg_witness_plan_create / zg_witness_run_dev): the recorded program
of zero_g's WnnCircuit replayed by the GPU must write, for every image, exactly the advice columns and public inputs
the host synthesis produces -- and create_proof from those device-resident columns must give the oracle's bytes."""
import ctypes

import numpy as np
import pytest
import torch  # (before the first HIP call of the process)

pytestmark = pytest.mark.gpu


def _d2h(ptr: int, nbytes: int) -> np.ndarray:
    hip = ctypes.CDLL("libamdhip64.so")
    out = np.zeros(nbytes // 8, np.uint64)
    assert hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2) == 0
    return out


@pytest.fixture(params=[-1, 0, 3], ids=["lds", "hbm-only", "lds-3KB"])
def lds_form(request, zg):
    """ZG_WITNESS_LDS: the default (live values register-allocated into LDS cells), 0 (every operand in HBM: round 2's form)
    and a 3-KB cell budget that forces all but a few hundred values to stay in HBM (the spill path the large models take)."""
    zg.tuning_set("ZG_WITNESS_LDS", request.param)
    yield request.param
    zg.tuning_set("ZG_WITNESS_LDS", -1)


@pytest.mark.parametrize("model", ["tiny", "medium"])
def test_device_witness_equals_host_synthesis(ctx, zg, orc, model, lds_form):
    import witness_tape
    import wnn_circuit
    import wnn_model

    k, name = {"tiny": wnn_model.MNIST_TINY, "medium": wnn_model.MNIST_MEDIUM}[model]
    wnn = wnn_model.load_checked_in(name)
    prog = witness_tape.trace(wnn, k)
    plan = zg.WitnessPlan(ctx, prog.arrays())
    info = plan.info()
    if lds_form == 0:
        assert info["lds_bytes"] == 0 and info["values_in_lds"] == 0 and info["hbm_levels"] == info["levels"]
    elif lds_form == -1 and model == "tiny":  # the whole live set fits: no operand in HBM, no barrier waits for a store
        assert info["values_in_hbm"] == 0 and info["hbm_levels"] == 0 and 0 < info["lds_bytes"] <= 160 * 1024
        assert info["values_in_lds"] > 18000 and info["narrow_cells"] < 6000 and info["wide_values"] < info["values_in_lds"] // 2
    else:  # medium at the full budget, or the 3-KB budget: both kinds of operands, some levels with the global barrier
        assert info["values_in_lds"] > 0 and info["values_in_hbm"] > 0 and 0 < info["hbm_levels"] <= info["levels"]
        assert info["lds_bytes"] <= 160 * 1024
    real = wnn_model.load_test_image()
    rng = np.random.default_rng(11)
    images = [real, np.zeros_like(real), np.full_like(real, 255)] + [rng.integers(0, 256, size=real.shape, dtype=real.dtype)
                                                                      for _ in range(2 if model == "tiny" else 0)]
    n = 1 << k
    bufs = [torch.full((6 * n * 4,), -1, dtype=torch.int64, device="cuda") for _ in images]  # (stale contents must go)
    inst = plan.run(np.stack(images), [b.data_ptr() for b in bufs])
    for im, buf, got_inst in zip(images, bufs, inst):
        cs, asg, ilen, scores = wnn_circuit.build(wnn, im, k)
        want = asg.advice_values()
        got = _d2h(buf.data_ptr(), want.nbytes).reshape(want.shape)
        assert np.array_equal(got, want)
        assert np.array_equal(got_inst, asg.instance_values(ilen)[0])
    plan.close()


def test_device_witness_of_the_k17_stand_in(ctx, zg):
    """model_49input_8192entry_4hash_6bpi's shape (seeded stand-in): 49-bit filter inputs, so the hash gadget cubes to
    147 bits and divides by a 53-bit modulus -- the wide multiplications and the division of the program's integer
    arithmetic.  360 000 operations, 215 levels; checked against the reference interpreter of the recorded program
    (itself checked against the host synthesis in tests/test_witness_tape.py) and against Wnn::predict."""
    import witness_tape
    import wnn_model
    from circuit import R

    k = 17
    wnn = wnn_model.synthetic_wnn()
    prog = witness_tape.trace(wnn, k)
    plan = zg.WitnessPlan(ctx, prog.arrays())
    real = wnn_model.load_test_image()
    images = [real, np.random.default_rng(2).integers(0, 256, size=real.shape, dtype=real.dtype)]
    n = 1 << k
    bufs = [torch.full((6 * n * 4,), -1, dtype=torch.int64, device="cuda") for _ in images]
    inst = plan.run(np.stack(images), [b.data_ptr() for b in bufs])
    mont = pow(2, 256, R)
    for im, buf, got_inst in zip(images, bufs, inst):
        adv, scores = prog.run(im)
        assert scores == wnn.predict(im)
        assert [zg.fr_to_int(x) for x in got_inst] == scores
        got = _d2h(buf.data_ptr(), 6 * n * 32).reshape(6, n, 4)
        want = np.zeros((6, n, 4), np.uint64)
        for c in range(6):
            for r in np.nonzero(np.array([v != 0 for v in adv[c]]))[0]:
                want[c, r] = zg.int_to_limbs(adv[c][r] * mont % R)
        assert np.array_equal(got, want)
    plan.close()


def test_proofs_from_device_witness(ctx, zg, orc):
    """image bytes -> advice columns in the prover's slots -> lock-step batch of proofs, nothing but the image and the
    class scores crossing PCIe: bytes == the oracle's create_proof of the host-synthesised witness."""
    import witness_tape
    import wnn_circuit
    import wnn_model

    orc.load().orc_set_threads(16)
    k, name = wnn_model.MNIST_TINY
    wnn = wnn_model.load_checked_in(name)
    real = wnn_model.load_test_image()
    rng = np.random.default_rng(3)
    images = [real] + [rng.integers(0, 256, size=real.shape, dtype=real.dtype) for _ in range(2)]
    built = [wnn_circuit.build(wnn, im, k) for im in images]
    cs, asg0, ilen, _ = built[0]
    img = cs.to_c()
    params = orc.params_new(k, 0x5EED)
    vk_repr = orc.fr_from_int(0xC0FFEE)
    pk = orc.ProvingKey(img, asg0.fixed_values(), asg0.sigma_values(), params, vk_repr)
    prover = zg.Prover(ctx, img, asg0.fixed_values(), asg0.sigma_values(), params.g_np(), params.g_lagrange_np(), vk_repr)
    prover.set_batch(3)
    prover.set_overlap(False)
    plan = zg.WitnessPlan(ctx, witness_tape.trace(wnn, k).arrays())
    for seeds in ([31, 32, 33], [41, 42, 43]):  # twice: the second run overwrites the blinding rows of the first
        inst = plan.run(np.stack(images), [prover.advice_slot(b) for b in range(3)])
        got, sts = prover.prove_batch(None, [i[None, :, :] for i in inst], seeds, device=True)
        assert sts == [0, 0, 0]
        for b, (_, asg, _, scores) in enumerate(built):
            st, want, _ = orc.create_proof(pk, asg.advice_values(), asg.instance_values(ilen), seeds[b])
            assert st == 0 and got[b] == want, f"proof {b}"
    # the same through the one-call form of Wnn::proof (image bytes -> proof bytes + outputs)
    got, outputs, sts = prover.prove_images(plan, np.stack(images), [51, 52, 53])
    assert sts == [0, 0, 0]
    for b, (_, asg, _, scores) in enumerate(built):
        st, want, _ = orc.create_proof(pk, asg.advice_values(), asg.instance_values(ilen), 51 + b)
        assert st == 0 and got[b] == want
        assert [zg.fr_to_int(x) for x in outputs[b]] == scores
        assert orc.verify_proof_pairing(pk, outputs[b][None, :, :], got[b]) == 1
    plan.close()
    prover.close()


def test_every_operation_on_wide_operands(ctx, zg, lds_form):
    """The circuit's own programs mostly move small integers; this one drives every opcode with random 256-bit
    operands, carries across all four words, shifts by 0..255 (and beyond, for the variable shift), divisors from 1
    to 2^64 - 1 and table reads on both sides of the bound -- against the reference interpreter."""
    import random

    import symint
    from circuit import R

    rnd = random.Random(20261004)
    t = symint.Tape()

    def raw_const(v):  # (Tape.const reduces modulo r: wide values go into the pool directly)
        t.consts.append(v)
        return t.emit("CONST", imm=len(t.consts) - 1)

    wide = [raw_const(rnd.getrandbits(256)) for _ in range(24)] + [raw_const(v) for v in (0, 1, (1 << 256) - 1, (1 << 64) - 1, 1 << 64, 1 << 255)]
    small = [raw_const(rnd.getrandbits(rnd.choice((1, 8, 31, 32, 33, 63)))) for _ in range(12)]
    px = [t.pixel(i) for i in range(4)]
    base = t.add_table([rnd.getrandbits(64) for _ in range(16)])
    out = []
    vals = lambda sym: t.consts[t.ops[sym.slot][3]]  # value of a CONST slot
    for _ in range(40):
        a, b = rnd.choice(wide), rnd.choice(wide)
        hi, lo = (a, b) if vals(a) >= vals(b) else (b, a)
        out += [a + b, hi - lo, a * b, a * rnd.choice(small), a + rnd.getrandbits(64), a * rnd.getrandbits(64)]
        out += [a >> rnd.randrange(256), a << rnd.randrange(256), a & rnd.getrandbits(64), a // rnd.choice((1, 2, 3, (1 << 64) - 1, rnd.getrandbits(64) | 1, rnd.getrandbits(33) | 1))]
        s_ = rnd.choice(small)
        out += [(rnd.getrandbits(64) | (1 << 63)) - s_]  # RSUBI: imm >= a
        out += [a >> rnd.choice(small + px), symint.gt(s_, vals(s_)), symint.gt(s_, max(0, vals(s_) - 1)), symint.ge(s_, vals(s_)),
                symint.ge(s_, vals(s_) + 1), symint.eq(s_, vals(s_)), symint.eq(a, vals(a) & symint.M64), symint.gt(a, symint.M64)]
    for idx in (0, 5, 15, 16, 1 << 40):  # in range, last, first outside, far outside
        c = raw_const(idx)
        out.append(t.emit("TABLE", c.slot, imm=base, deps=(c.slot,)))
    out += [p * 3 + 1 for p in px]
    # one advice column of 2^k rows showing the results; instance = the first four
    k = (len(out) - 1).bit_length()
    prog_cells = {(0, i): o.slot for i, o in enumerate(out)}
    import witness_tape

    prog = witness_tape.WitnessProgram(t, prog_cells, {i: out[i].slot for i in range(4)}, 1, k, 4)
    plan = zg.WitnessPlan(ctx, prog.arrays())
    image = np.array([[0, 7, 200, 255]], dtype=np.uint8)
    buf = torch.full(((1 << k) * 4,), -1, dtype=torch.int64, device="cuda")
    inst = plan.run(image, [buf.data_ptr()])
    v = t.run(image.reshape(-1))
    got = _d2h(buf.data_ptr(), (1 << k) * 32).reshape(1 << k, 4)
    for i, o in enumerate(out):
        assert zg.fr_to_int(got[i]) == v[o.slot] % R, f"result {i}: op {symint.OPS[t.ops[o.slot][0]]}"
    assert not got[len(out):].any()
    assert [zg.fr_to_int(x) for x in inst[0]] == [v[out[i].slot] % R for i in range(4)]
    plan.close()


def _run_wrapping(t, image_bytes):
    """symint.Tape.run without the tape contract's no-underflow assertions: the device computes modulo 2^256 throughout"""
    import symint

    M = symint.M256
    v = [0] * len(t.ops)
    for i, (op, a, b, imm) in enumerate(t.ops):
        name = symint.OPS[op]
        r = {"CONST": lambda: t.consts[imm], "PIXEL": lambda: int(image_bytes[imm]), "ADD": lambda: v[a] + v[b], "SUB": lambda: v[a] - v[b],
             "MUL": lambda: v[a] * v[b], "ADDI": lambda: v[a] + imm, "RSUBI": lambda: imm - v[a], "MULI": lambda: v[a] * imm,
             "SHRI": lambda: v[a] >> imm, "SHLI": lambda: v[a] << imm, "ANDI": lambda: v[a] & imm,
             "SHRV": lambda: v[a] >> v[b] if v[b] < 256 else 0, "GTI": lambda: int(v[a] > imm), "GEI": lambda: int(v[a] >= imm),
             "EQI": lambda: int(v[a] == imm), "DIVI": lambda: v[a] // imm,
             "TABLE": lambda: t.table[imm + v[a]] if imm + v[a] < len(t.table) else 0}[name]()
        v[i] = r & M
    return v


def test_chained_operations_through_recycled_lds_cells(ctx, zg, lds_form):
    """The LDS form's own risks: a value read from the cell another value lived in a level earlier, an 8-byte cell for a
    value whose bound was wrong, a wide value in the narrow class.  A random DAG of 60 levels x 96 operations, every operand
    from an EARLIER level (mostly the previous one, some far back: cells recycle constantly), narrow and wide values mixed,
    subtractions that wrap, shifts back into 64 bits, comparisons and masks that make wide values narrow again -- four images,
    against the reference interpreter, with every slot shown in an advice cell."""
    import random

    import symint
    import witness_tape
    from circuit import R

    rnd = random.Random(555)
    t = symint.Tape()

    def raw_const(v):
        t.consts.append(v)
        return t.emit("CONST", imm=len(t.consts) - 1)

    px = [t.pixel(i) for i in range(16)]
    base = t.add_table([rnd.getrandbits(64) for _ in range(64)])
    levels = [px + [raw_const(rnd.getrandbits(b)) for b in (1, 7, 20, 33, 63, 64, 65, 128, 200, 256)] + [raw_const(v) for v in (0, 1, R - 1, (1 << 64) - 1)]]
    for lv in range(60):
        cur = []
        for _ in range(96):
            src = levels[-1] if rnd.random() < 0.7 else levels[rnd.randrange(len(levels))]
            a, b = rnd.choice(src), rnd.choice(levels[rnd.randrange(len(levels))])
            kind = rnd.randrange(14)
            if kind == 0: r = a + b
            elif kind == 1: r = a - b  # (wraps modulo 2^256 when b > a: the bound must call it wide)
            elif kind == 2: r = a * b
            elif kind == 3: r = a + rnd.getrandbits(rnd.choice((3, 30, 64)))
            elif kind == 4: r = a * rnd.getrandbits(rnd.choice((2, 16, 40)))
            elif kind == 5: r = a >> rnd.choice((0, 1, 13, 64, 100, 192, 200, 255))
            elif kind == 6: r = (a & symint.M64) << rnd.choice((0, 1, 5, 40, 70))
            elif kind == 7: r = a & rnd.getrandbits(rnd.choice((1, 8, 32, 64)))
            elif kind == 8: r = a >> (b & 255)
            elif kind == 9: r = symint.gt(a, rnd.getrandbits(20))
            elif kind == 10: r = a // (rnd.getrandbits(rnd.choice((3, 20, 53))) | 1)
            elif kind == 11: r = (rnd.getrandbits(64)) - (a & rnd.getrandbits(30))  # RSUBI, may or may not wrap
            elif kind == 12:
                idx = a & 63
                r = t.emit("TABLE", idx.slot, imm=base, deps=(idx.slot,))
            else: r = symint.eq(a & 3, 1)
            cur.append(r)
        levels.append(cur)
    out = [v for lv in levels for v in lv]
    k = (len(out) - 1).bit_length()
    prog = witness_tape.WitnessProgram(t, {(0, i): o.slot for i, o in enumerate(out)}, {i: out[-1 - i].slot for i in range(4)}, 1, k, 16)
    plan = zg.WitnessPlan(ctx, prog.arrays())
    info = plan.info()
    if lds_form != 0:
        assert info["values_in_lds"] > 1000 and info["wide_cells"] > 0 and info["narrow_cells"] > 0
        # cells are recycled: far fewer cells than values that lived in one
        assert info["narrow_cells"] + info["wide_cells"] < info["values_in_lds"] // 3
    images = np.array([[0] * 16, [255] * 16, list(range(0, 256, 16)), [rnd.randrange(256) for _ in range(16)]], dtype=np.uint8)
    bufs = [torch.full(((1 << k) * 4,), -1, dtype=torch.int64, device="cuda") for _ in images]
    inst = plan.run(images, [b.data_ptr() for b in bufs])
    for im, buf, got_inst in zip(images, bufs, inst):
        v = _run_wrapping(t, im.reshape(-1))
        got = _d2h(buf.data_ptr(), (1 << k) * 32).reshape(1 << k, 4)
        for i, o in enumerate(out):
            assert zg.fr_to_int(got[i]) == v[o.slot] % R, f"slot {o.slot} (level {i // 96}): op {symint.OPS[t.ops[o.slot][0]]}"
        assert [zg.fr_to_int(x) for x in got_inst] == [v[out[-1 - i].slot] % R for i in range(4)]
    plan.close()


def test_prove_images_refuses_more_images_than_slots(ctx, zg, orc):
    from circuits import toy_circuit

    cs, asg, ilen = toy_circuit(6)
    params = orc.params_new(6, 0xABCDEF)
    prover = zg.Prover(ctx, cs.to_c(), asg.fixed_values(), asg.sigma_values(), params.g_np(), params.g_lagrange_np(), orc.fr_from_int(1))
    prover.set_batch(2)
    plan = zg.WitnessPlan(ctx, dict(ops=np.array([[1, 0, 0, 0]], dtype=np.uint64), level_start=np.array([0, 1], dtype=np.uint32),
                                    consts=np.zeros((1, 4), np.uint64), table=np.zeros(1, np.uint64),
                                    cell_slot=np.full((cs.n_advice, 64), 0xFFFFFFFF, np.uint32),
                                    instance_slots=np.zeros(0, np.uint32), image_bytes=1))
    with pytest.raises(zg.ZgError):
        prover.prove_images(plan, np.zeros((3, 1), np.uint8), [1, 2, 3])
    plan.close()
    prover.close()


def test_plan_rejects_malformed_programs(ctx, zg):
    """The program runs on the GPU unchecked, so zg_witness_plan_create refuses everything it can see statically."""
    def arrays(**over):
        a = dict(ops=np.array([[1, 0, 0, 0], [5, 0, 0, 7]], dtype=np.uint64),  # PIXEL 0; ADDI slot0 + 7
                 level_start=np.array([0, 1, 2], dtype=np.uint32), consts=np.zeros((1, 4), np.uint64),
                 table=np.zeros(1, np.uint64), cell_slot=np.full((1, 4), 0xFFFFFFFF, np.uint32),
                 instance_slots=np.array([1], dtype=np.uint32), image_bytes=2)
        a.update(over)
        return a

    ok = zg.WitnessPlan(ctx, arrays())
    buf = torch.zeros(4 * 4, dtype=torch.int64, device="cuda")
    inst = ok.run(np.array([[5, 9]], dtype=np.uint8), [buf.data_ptr()])
    assert zg.fr_to_int(inst[0, 0]) == 12
    ok.close()
    bad = [
        dict(ops=np.array([[1, 0, 0, 0], [5, 1, 0, 7]], dtype=np.uint64)),              # reads its own slot
        dict(ops=np.array([[1, 0, 0, 2], [5, 0, 0, 7]], dtype=np.uint64)),              # pixel outside the image
        dict(ops=np.array([[0, 0, 0, 3], [5, 0, 0, 7]], dtype=np.uint64)),              # constant outside the pool
        dict(ops=np.array([[1, 0, 0, 0], [16, 0, 0, 1]], dtype=np.uint64)),             # table base outside the table
        dict(ops=np.array([[1, 0, 0, 0], [15, 0, 0, 0]], dtype=np.uint64)),             # division by zero
        dict(ops=np.array([[1, 0, 0, 0], [99, 0, 0, 0]], dtype=np.uint64)),             # unknown opcode
        dict(level_start=np.array([0, 2, 2], dtype=np.uint32)),                         # operand in the same level
        dict(cell_slot=np.full((1, 4), 2, np.uint32)),                                  # cell shows a missing slot
        dict(instance_slots=np.array([2], dtype=np.uint32)),
        # trailing levels that hold no operation are walked by the kernel all the same: a level that
        # reaches beyond the operations, and a tail that runs backwards
        dict(ops=np.array([[1, 0, 0, 0], [1, 0, 0, 1], [0, 0, 0, 0]], dtype=np.uint64), instance_slots=np.array([2], dtype=np.uint32),
             level_start=np.array([0, 3, 1000000, 3], dtype=np.uint32)),
        dict(level_start=np.array([0, 1, 2, 1, 2], dtype=np.uint32)),
        dict(level_start=np.array([0, 1, 3, 2], dtype=np.uint32)),
    ]
    for over in bad:
        with pytest.raises(zg.ZgError):
            zg.WitnessPlan(ctx, arrays(**over))


def test_prove_images_refuses_a_plan_of_another_shape(ctx, zg, orc):
    """zg_prover_prove_images writes the plan's [n_advice][2^k] columns into the prover's slots: a plan recorded for
    another circuit shape (more columns, more rows) must be refused, not overrun the slot."""
    from circuits import toy_circuit

    cs, asg, ilen = toy_circuit(6)
    params = orc.params_new(6, 0xABCDEF)
    prover = zg.Prover(ctx, cs.to_c(), asg.fixed_values(), asg.sigma_values(), params.g_np(), params.g_lagrange_np(), orc.fr_from_int(5))

    def plan(n_advice, k):
        return zg.WitnessPlan(ctx, dict(ops=np.array([[1, 0, 0, 0], [5, 0, 0, 7]], dtype=np.uint64),
                                        level_start=np.array([0, 1, 2], dtype=np.uint32), consts=np.zeros((1, 4), np.uint64),
                                        table=np.zeros(1, np.uint64), cell_slot=np.full((n_advice, 1 << k), 0xFFFFFFFF, np.uint32),
                                        instance_slots=np.array([1], dtype=np.uint32), image_bytes=2))

    img = np.array([[5, 9]], dtype=np.uint8)
    for n_advice, k in ((cs.n_advice + 1, 6), (cs.n_advice, 7), (cs.n_advice, 5)):
        bad = plan(n_advice, k)
        with pytest.raises(zg.ZgError) as e:
            prover.prove_images(bad, img, [1])
        assert e.value.status == -1
        bad.close()
    # the matching shape goes through (the all-zero witness with instance 12 is no valid statement: bytes, not validity)
    ok = plan(cs.n_advice, 6)
    proofs, outputs, sts = prover.prove_images(ok, img, [1], raise_on_error=False)
    assert zg.fr_to_int(outputs[0, 0]) == 12
    ok.close()
    # ... and the prover still proves the real witness afterwards
    pk = orc.ProvingKey(cs.to_c(), asg.fixed_values(), asg.sigma_values(), params, orc.fr_from_int(5))
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    assert prover.prove(adv, inst, 3) == orc.create_proof(pk, adv, inst, 3)[1]
    prover.close()
