"""GPU parity: HIP Pippenger MSM (through the C ABI) == oracle restatement of halo2's best_multiexp,
compared as normalised points (bit-exact), on uniform and on advice-like sparse scalars."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def srs(orc):
    prm = orc.params_new(12)
    return prm.g_np(), prm.g_lagrange_np()


@pytest.mark.parametrize("c", [0, 4, 8, 11, 13, 16])
def test_msm_matches_oracle_uniform(ctx, zg, orc, srs, c):
    g, _ = srs
    n = g.shape[0]
    bases = ctx.register_bases(g, c)
    s = orc.fill_fr(21, n)
    assert np.array_equal(ctx.msm(bases, s), orc.msm(s, g, threads=8))
    bases.free()


def test_msm_sparse_and_edge_scalars(ctx, zg, orc, srs):
    _, gl = srs
    n = gl.shape[0]
    bases = ctx.register_bases(gl)
    sp = orc.fill_fr_sparse(5, n)
    assert np.array_equal(ctx.msm(bases, sp), orc.msm(sp, gl, threads=8))
    # all zeros -> identity (0, 1, 0)
    z = np.zeros((n, 4), np.uint64)
    r = ctx.msm(bases, z)
    assert not r[:4].any() and not r[8:].any() and zg.fq_to_int(r[4:8]) == 1
    # unit vector -> that base; r-1 (= -1) on one point -> its negation
    for idx in (0, 1, n - 1):
        e = np.zeros((n, 4), np.uint64)
        e[idx] = orc.fr_from_int(1)
        r = ctx.msm(bases, e)
        assert np.array_equal(r[:8], gl[idx]) and zg.fq_to_int(r[8:]) == 1
    e = np.zeros((n, 4), np.uint64)
    e[7] = orc.fr_from_int(zg.FR_MODULUS - 1)
    r = ctx.msm(bases, e)
    assert np.array_equal(r[:4], gl[7][:4])
    assert zg.fq_to_int(r[4:8]) == zg.FQ_MODULUS - zg.fq_to_int(gl[7][4:])
    # all ones: one hot bucket holding every point
    ones = np.tile(orc.fr_from_int(1), (n, 1))
    assert np.array_equal(ctx.msm(bases, ones), orc.msm(ones, gl, threads=8))
    # maximal scalars r-1 everywhere (top window + carry path)
    mx = np.tile(orc.fr_from_int(zg.FR_MODULUS - 1), (n, 1))
    assert np.array_equal(ctx.msm(bases, mx), orc.msm(mx, gl, threads=8))
    # ragged: fewer scalars than bases, and n = 1
    for m in (1, 2, 3, 100, n - 1):
        s = orc.fill_fr(900 + m, m)
        assert np.array_equal(ctx.msm(bases, s), orc.msm(s, gl[:m], threads=4))
    # empty
    r = ctx.msm(bases, np.zeros((0, 4), np.uint64))
    assert not r[:4].any() and not r[8:].any()
    bases.free()


def test_msm_batch_matches_single(ctx, zg, orc, srs):
    g, _ = srs
    n = g.shape[0]
    bases = ctx.register_bases(g)
    batch = 6
    s = np.stack([orc.fill_fr(31 + b, n) if b % 2 == 0 else orc.fill_fr_sparse(31 + b, n) for b in range(batch)])
    got = ctx.msm_batch(bases, s)
    for b in range(batch):
        assert np.array_equal(got[b], orc.msm(s[b], g, threads=8)), b
    bases.free()


def test_msm_linearity_full_size(ctx, zg, orc):
    """k = 14 (BASELINE configs[1]): MSM(a) + MSM(b) == MSM(a + b), and parity with the oracle."""
    prm = orc.params_new(14)
    g = prm.g_np()
    n = g.shape[0]
    bases = ctx.register_bases(g)
    a, b = orc.fill_fr(1, n), orc.fill_fr_sparse(2, n)
    L = orc.load()
    import ctypes

    ab = np.zeros_like(a)
    for i in range(n):
        L.orc_fr_add(ctypes.c_void_p(ab[i].ctypes.data), ctypes.c_void_p(a[i].ctypes.data),
                     ctypes.c_void_p(b[i].ctypes.data))
    ra, rb, rab = ctx.msm(bases, a), ctx.msm(bases, b), ctx.msm(bases, ab)
    assert np.array_equal(zg.g1_sum(np.stack([ra, rb])), rab)
    assert np.array_equal(ra, orc.msm(a, g, threads=8))
    assert np.array_equal(rb, orc.msm(b, g, threads=8))
    bases.free()


def test_msm_argument_errors(ctx, zg, orc, srs):
    g, _ = srs
    bases = ctx.register_bases(g[:16])
    with pytest.raises(zg.ZgError) as e:
        ctx.msm(bases, orc.fill_fr(1, 17))
    assert e.value.status == -1
    with pytest.raises(zg.ZgError):
        ctx.register_bases(g[:16], 17)
    bases.free()


def test_params_new_matches_oracle(ctx, zg, orc):
    """ParamsKZG::new(k) on the GPU == the oracle's setup for the same toxic scalar."""
    k = 7
    s = orc.fill_fr(0x5EED, 1)[0]
    g, gl = ctx.params_new(k, s)
    prm = orc.params_new(k, 0x5EED)
    assert np.array_equal(g, prm.g_np())
    assert np.array_equal(gl, prm.g_lagrange_np())


@pytest.mark.parametrize("latency", [True, False])
def test_both_reduction_forms_match_the_oracle(ctx, zg, orc, srs, latency):
    """zg_ctx_set_msm_latency: two lanes per addition (default) and one lane per addition."""
    g, gl = srs
    n = gl.shape[0]
    ctx.set_msm_latency(latency)
    try:
        for c in (0, 6, 12):
            bases = ctx.register_bases(gl, c)
            for s in (orc.fill_fr(31, n), orc.fill_fr_sparse(32, n), np.tile(orc.fr_from_int(3), (n, 1))):
                assert np.array_equal(ctx.msm(bases, s), orc.msm(s, gl, threads=8))
            e = np.zeros((n, 4), np.uint64)
            e[5] = orc.fr_from_int(1)
            assert np.array_equal(ctx.msm(bases, e)[:8], gl[5])
            assert not ctx.msm(bases, np.zeros((n, 4), np.uint64))[8:].any()
            bases.free()
    finally:
        ctx.set_msm_latency(True)


@pytest.mark.parametrize("w", [3, 9, 13, 16])
def test_free_position_digits_on_adversarial_scalars(zg, orc, srs, w):
    """zg_bases_enable_bit_table: odd signed digits of w bits at free positions against the bit-position table, on a
    throughput-form context.  The recoding walks the scalar with a 64-bit sliding register and folds a negative digit's
    carry into it: the vectors below put runs of ones, alternating bits, single bits and the extreme values across every
    32-bit limb boundary -- each must give the oracle's point."""
    _, gl = srs
    n = gl.shape[0]
    c2 = zg.Ctx(0)
    c2.set_msm_latency(False)
    bases = c2.register_bases(gl)
    c2.enable_bit_table(bases, w)
    R = zg.FR_MODULUS
    special = [0, 1, 2, 3, R - 1, R - 2, (R - 1) // 2, (R + 1) // 2, (1 << 253), (1 << 253) - 1, (1 << 253) + 1]
    special += [1 << e for e in (28, 29, 30, 31, 32, 33, 47, 48, 63, 64, 65, 95, 96, 127, 128, 191, 192, 224, 250, 252)]
    special += [(1 << e) - 1 for e in (15, 16, 17, 31, 32, 33, 63, 64, 65, 128, 200, 253)]
    special += [int("aa" * 31, 16), int("55" * 31, 16), int("ff" * 31, 16) % R, int("0f" * 31, 16), int("f0" * 31, 16) % R]
    special += [((1 << w) - 1) << s for s in (0, 1, 17, 31, 32, 33, 48, 64 - w, 64, 200)]
    special += [(((1 << (w - 1)) + 1) << s) % R for s in (0, 15, 31, 32, 63, 64, 100, 230)]
    special = [v % R for v in special]
    rng = np.random.default_rng(w)
    for trial in range(2):
        s = orc.fill_fr(700 + 10 * w + trial, n)
        pos = rng.permutation(n)[: len(special)]
        for p_, v in zip(pos, special):
            s[p_] = orc.fr_from_int(v)
        assert np.array_equal(c2.msm(bases, s), orc.msm(s, gl, threads=8)), f"trial {trial}"
    only = np.zeros((n, 4), np.uint64)  # the special values alone (no random mass to hide behind), one per point
    for i, v in enumerate(special[:n]):
        only[i] = orc.fr_from_int(v)
    assert np.array_equal(c2.msm(bases, only), orc.msm(only, gl, threads=8))
    mx = np.tile(orc.fr_from_int(R - 1), (n, 1))  # every scalar r - 1: one bucket pattern, all carries
    assert np.array_equal(c2.msm(bases, mx), orc.msm(mx, gl, threads=8))
    sp = orc.fill_fr_sparse(5, n)
    assert np.array_equal(c2.msm(bases, sp), orc.msm(sp, gl, threads=8))
    # a latency-form context on the same base set keeps the window table and gives the same point
    c3 = zg.Ctx(0)
    assert np.array_equal(c3.msm(bases, only), orc.msm(only, gl, threads=8))
    c3.close()
    with pytest.raises(zg.ZgError):
        c2.enable_bit_table(bases, w + 1 if w < 16 else 15)  # another width: refused
    bases.free()
    c2.close()


@pytest.mark.parametrize("c", [0, 4, 7, 9])
def test_digit_tables_give_the_same_points(ctx, zg, orc, srs, c):
    """zg_bases_enable_digit_table: the latency form without buckets (every multiple of every window; a flat sum of
    gathered points folded by a tree) == oracle, on uniform, sparse and adversarial scalars, ragged lengths, batches."""
    g, gl = srs
    n = gl.shape[0]
    bases = ctx.register_bases(gl)
    ctx.set_msm_latency(True)
    ctx.enable_digit_table(bases, c)
    R = zg.FR_MODULUS
    vectors = [orc.fill_fr(41, n), orc.fill_fr_sparse(42, n), np.zeros((n, 4), np.uint64),
               np.tile(orc.fr_from_int(1), (n, 1)), np.tile(orc.fr_from_int(R - 1), (n, 1)),
               np.tile(orc.fr_from_int((1 << 253) + 12345), (n, 1))]
    # digit boundaries: runs of ones, half-window values (the signed recoding's carry), powers of two
    edge = np.stack([orc.fr_from_int(v % R) for v in
                     ([(1 << b) - 1 for b in range(1, 255, 5)] + [1 << b for b in range(0, 254, 7)] +
                      [(1 << (9 * j + 8)) for j in range(20)] + [R - 1 - (1 << b) for b in range(0, 250, 11)])])
    e = np.zeros((n, 4), np.uint64)
    e[: len(edge)] = edge
    vectors.append(e)
    got = ctx.msm_batch(bases, np.stack(vectors))
    for b, s in enumerate(vectors):
        assert np.array_equal(got[b], orc.msm(s, gl, threads=8)), (c, b)
    for m in (1, 2, 63, 64, 65, 1000, n - 1):  # fewer scalars than bases
        s = orc.fill_fr(700 + m, m)
        assert np.array_equal(ctx.msm(bases, s), orc.msm(s, gl[:m], threads=4)), (c, m)
    r = ctx.msm(bases, np.zeros((0, 4), np.uint64))
    assert not r[:4].any() and not r[8:].any()
    bases.free()


@pytest.mark.parametrize("rounds", [1, 2, 3, 4])
@pytest.mark.parametrize("bit_table", [False, True])
def test_batched_affine_rounds_give_the_same_points(zg, orc, srs, rounds, bit_table):
    """ZG_MSM_AFFINE: R rounds of pairwise AFFINE additions (one inversion per launch and round, shared through HBM by
    Montgomery's trick) before the buckets' XYZZ chains, throughput form.  An affine chord has no formula for P + P,
    P + (-P) or an identity operand, and ONE zero denominator would poison every sum of the launch -- so the base sets
    here are built to produce all of them inside the paired-up bucket order: repeated bases (doublings when two equal
    points share a bucket), a base and its negation (cancelling pairs), identity bases, and scalars that put them into the
    same bucket (equal scalars, +-s).  Every result == the oracle's best_multiexp."""
    _, gl = srs
    n = 1 << 10
    base = gl[:n].copy()
    FQ = zg.FQ_MODULUS

    def neg(pt):
        out = pt.copy()
        out[4:] = zg.fq_from_int((FQ - zg.fq_to_int(pt[4:])) % FQ)  # (x, -y)
        return out

    adv = base.copy()
    adv[1] = adv[0]                      # P, P        -> a doubling when the scalars agree
    adv[3] = neg(adv[2])                 # P, -P       -> cancels when the scalars agree
    adv[4:8] = 0                         # identities among the bases
    for i in range(16, 48, 2):           # sixteen equal points in a row: a bucket of doublings, round after round
        adv[i] = adv[16]
    for i in range(48, 64):              # alternating P, -P
        adv[i] = adv[48] if i % 2 == 0 else neg(adv[48])
    c2 = zg.Ctx(0)
    c2.set_msm_latency(False)
    before = zg.tuning_get("ZG_MSM_AFFINE")
    zg.tuning_set("ZG_MSM_AFFINE", rounds)
    try:
        for bases_np in (base, adv):
            bases = c2.register_bases(bases_np)
            if bit_table:
                c2.enable_bit_table(bases, 9)
            R = zg.FR_MODULUS
            same = np.tile(orc.fr_from_int(0x1234567), (n, 1))            # every point in the same buckets
            pm = same.copy()
            pm[1::2] = orc.fr_from_int(R - 0x1234567)                     # s, -s alternating: same bucket, opposite signs
            vectors = [orc.fill_fr(51, n), orc.fill_fr_sparse(52, n), np.zeros((n, 4), np.uint64), same, pm,
                       np.tile(orc.fr_from_int(1), (n, 1)), np.tile(orc.fr_from_int(R - 1), (n, 1))]
            small = np.zeros((n, 4), np.uint64)
            small[:64] = np.stack([orc.fr_from_int(v) for v in list(range(1, 33)) * 2])  # few entries: mostly pad pairs
            vectors.append(small)
            got = c2.msm_batch(bases, np.stack(vectors))
            for b, sc in enumerate(vectors):
                assert np.array_equal(got[b], orc.msm(sc, bases_np, threads=8)), (rounds, bit_table, b)
            for m in (1, 2, 3, 65, n - 1):  # ragged lengths
                sc = orc.fill_fr(800 + m, m)
                assert np.array_equal(c2.msm(bases, sc), orc.msm(sc, bases_np[:m], threads=4)), (rounds, m)
            bases.free()
    finally:
        zg.tuning_set("ZG_MSM_AFFINE", before)
        c2.close()


@pytest.mark.gpu
def test_field_inv_returns_zero_when_its_budget_runs_out_device_path():
    """Field::inv of a multiple of the modulus on the DEVICE: the loop ends (iteration budget) and the result is 0
    (tests/abi/field_probe.hip; one run, its own process)."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "abi", "field_probe")
    r = subprocess.run([exe, "device"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
