"""Pins the oracle (oracle/*.c) against values re-derived with Python big integers and against the
known-answer material the reference's own tests/doc hold for this path.

The reference has NO golden vector for MSM/NTT/commitment bytes (its tests only check that a proof
verifies: src/lib.rs:10-33, test_cli.sh:62-82), so what is pinned here is the arithmetic the path is
built from: moduli, Montgomery constants, 2G, r*G = inf, ROOT_OF_UNITY / DELTA / ZETA, and the
algebraic identities MSM and NTT must satisfy.  Parity with real halo2 bytes stays "unpinned".
"""
import ctypes
import random

import numpy as np

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
MONT = 1 << 256


def limbs(x):
    return np.array([(x >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)


def to_int(a):
    return sum(int(a[i]) << (64 * i) for i in range(4))


def p(a):
    return ctypes.c_void_p(a.ctypes.data)


def test_moduli_and_constants(orc):
    L = orc.load()
    fr_mod = (ctypes.c_uint64 * 4).in_dll(L, "ORC_FR_MODULUS")
    fq_mod = (ctypes.c_uint64 * 4).in_dll(L, "ORC_FQ_MODULUS")
    assert to_int(fr_mod) == R and to_int(fq_mod) == Q
    one_r = (ctypes.c_uint64 * 4).in_dll(L, "ORC_FR_ONE")
    one_q = (ctypes.c_uint64 * 4).in_dll(L, "ORC_FQ_ONE")
    assert to_int(one_r) == MONT % R and to_int(one_q) == MONT % Q
    # halo2curves bn256::Fr constants: GENERATOR = 7, S = 28
    rou = pow(7, (R - 1) >> 28, R)
    assert pow(rou, 1 << 28, R) == 1 and pow(rou, 1 << 27, R) != 1
    assert to_int((ctypes.c_uint64 * 4).in_dll(L, "ORC_FR_ROOT_OF_UNITY_RAW")) == rou
    assert to_int((ctypes.c_uint64 * 4).in_dll(L, "ORC_FR_ROOT_OF_UNITY")) == rou * MONT % R
    delta = pow(7, 1 << 28, R)
    assert to_int((ctypes.c_uint64 * 4).in_dll(L, "ORC_FR_DELTA_RAW")) == delta
    assert to_int((ctypes.c_uint64 * 4).in_dll(L, "ORC_FR_DELTA")) == delta * MONT % R
    zeta = pow(7, (R - 1) // 3, R)
    assert pow(zeta, 3, R) == 1 and zeta != 1
    assert to_int((ctypes.c_uint64 * 4).in_dll(L, "ORC_FR_ZETA_RAW")) == zeta
    assert to_int((ctypes.c_uint64 * 4).in_dll(L, "ORC_FR_ZETA")) == zeta * MONT % R


def test_field_ops_vs_bigint(orc):
    L = orc.load()
    rng = random.Random(1)
    for mod, nm in ((R, "fr"), (Q, "fq")):
        f = lambda name: getattr(L, f"orc_{nm}_{name}")
        for _ in range(300):
            x, y = rng.randrange(mod), rng.randrange(mod)
            xm, ym, zm, z = (np.zeros(4, np.uint64) for _ in range(4))
            f("from_raw")(p(xm), p(limbs(x)))
            f("from_raw")(p(ym), p(limbs(y)))
            assert to_int(xm) == x * MONT % mod
            for op, want in (("mul", x * y % mod), ("add", (x + y) % mod), ("sub", (x - y) % mod)):
                f(op)(p(zm), p(xm), p(ym))
                f("to_raw")(p(z), p(zm))
                assert to_int(z) == want, (nm, op)
            f("inv")(p(zm), p(xm))
            f("to_raw")(p(z), p(zm))
            assert to_int(z) == pow(x, -1, mod)
            f("neg")(p(zm), p(xm))
            f("to_raw")(p(z), p(zm))
            assert to_int(z) == (-x) % mod
        # edge cases
        for x, y in ((0, 0), (mod - 1, mod - 1), (1, mod - 1), (0, 5)):
            xm, ym, zm, z = (np.zeros(4, np.uint64) for _ in range(4))
            f("from_raw")(p(xm), p(limbs(x)))
            f("from_raw")(p(ym), p(limbs(y)))
            f("mul")(p(zm), p(xm), p(ym))
            f("to_raw")(p(z), p(zm))
            assert to_int(z) == x * y % mod
            f("add")(p(zm), p(xm), p(ym))
            f("to_raw")(p(z), p(zm))
            assert to_int(z) == (x + y) % mod


def test_challenge_reduction(orc):
    """EvmTranscript squeezes a 256-bit keccak output and reduces it mod r."""
    L = orc.load()
    rng = random.Random(7)
    for v in [0, R - 1, R, R + 1, 2**256 - 1] + [rng.randrange(2**256) for _ in range(50)]:
        b = np.frombuffer(v.to_bytes(32, "big"), dtype=np.uint8).copy()
        out, raw = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
        L.orc_fr_from_be_bytes_reduce(p(out), p(b))
        L.orc_fr_to_raw(p(raw), p(out))
        assert to_int(raw) == v % R


def test_g1_known_answers(orc):
    L = orc.load()
    g, d, a = np.zeros(12, np.uint64), np.zeros(12, np.uint64), np.zeros(8, np.uint64)
    L.orc_g1_generator(p(g))
    L.orc_g1_double(p(d), p(g))
    L.orc_g1_to_affine(p(a), p(d))
    # 2*G on y^2 = x^3 + 3 from G = (1, 2): tangent slope 3/4
    lam = 3 * pow(4, -1, Q) % Q
    x2 = (lam * lam - 2) % Q
    y2 = (lam * (1 - x2) - 2) % Q
    assert x2 == 0x030644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD3
    assert y2 == 0x15ED738C0E0A7C92E7845F96B2AE9C0A68A6A449E3538FC7FF3EBF7A5A18A2C4
    assert orc.fq_to_int(a[:4]) == x2 and orc.fq_to_int(a[4:]) == y2
    assert L.orc_g1a_on_curve(p(a)) == 1
    # (r-1)*G + G = identity
    k = np.zeros(4, np.uint64)
    L.orc_fr_from_raw(p(k), p(limbs(R - 1)))
    o, o2 = np.zeros(12, np.uint64), np.zeros(12, np.uint64)
    L.orc_g1_mul(p(o), p(g), p(k))
    L.orc_g1_add(p(o2), p(o), p(g))
    assert L.orc_g1_is_identity(p(o2)) == 1
    # mixed add == full add, add(P,P) == double, P + (-P) = identity
    L.orc_g1_add(p(o), p(d), p(g))
    ga = np.zeros(8, np.uint64)
    L.orc_g1_to_affine(p(ga), p(g))
    L.orc_g1_add_mixed(p(o2), p(d), p(ga))
    assert L.orc_g1_eq(p(o), p(o2)) == 1
    L.orc_g1_add(p(o), p(g), p(g))
    assert L.orc_g1_eq(p(o), p(d)) == 1
    n = np.zeros(12, np.uint64)
    L.orc_g1_neg(p(n), p(g))
    L.orc_g1_add(p(o), p(g), p(n))
    assert L.orc_g1_is_identity(p(o)) == 1


def test_fft_matches_definition_and_roundtrip(orc):
    for k in (1, 3, 6):
        d = orc.domain(6, k)
        a = orc.fill_fr(100 + k, 1 << k)
        f = orc.fft(a, d.fe("omega"))
        assert np.array_equal(f, orc.dft_naive(a, d.fe("omega")))
        # python big-int DFT of the first outputs
        om = orc.fr_to_int(d.fe("omega"))
        ai = [orc.fr_to_int(x) for x in a]
        for kk in range(min(4, 1 << k)):
            want = sum(ai[j] * pow(om, j * kk, R) for j in range(1 << k)) % R
            assert orc.fr_to_int(f[kk]) == want
        assert np.array_equal(orc.lagrange_to_coeff(d, f), a)


def test_domain_constants(orc):
    d = orc.domain(6, 14)  # cs.degree() = 6 -> quotient degree 5 -> extended_k = k + 3
    assert (d.k, d.extended_k, d.n, d.extended_n, d.quotient_poly_degree) == (14, 17, 1 << 14, 1 << 17, 5)
    om, eom = orc.fr_to_int(d.fe("omega")), orc.fr_to_int(d.fe("extended_omega"))
    assert pow(om, 1 << 14, R) == 1 and pow(om, 1 << 13, R) != 1
    assert pow(eom, 8, R) == om
    assert eom == pow(pow(7, (R - 1) >> 28, R), 1 << (28 - 17), R)
    assert orc.fr_to_int(d.fe("ifft_divisor")) == pow(1 << 14, -1, R)
    assert d.t_len == 8


def test_extended_roundtrip_and_coset_definition(orc):
    k = 4
    d = orc.domain(6, k)
    a = orc.fill_fr(5, 1 << k)
    ext = orc.coeff_to_extended(d, a)
    zeta = pow(7, (R - 1) // 3, R)
    eom = orc.fr_to_int(d.fe("extended_omega"))
    ai = [orc.fr_to_int(x) for x in a]
    for i in (0, 1, 5, (1 << d.extended_k) - 1):
        x = zeta * pow(eom, i, R) % R
        want = sum(c * pow(x, j, R) for j, c in enumerate(ai)) % R
        assert orc.fr_to_int(ext[i]) == want  # evaluation on the zeta-coset of the extended domain
    back = orc.extended_to_coeff(d, ext)
    assert back.shape[0] == 5 << k
    assert np.array_equal(back[: 1 << k], a)
    assert not back[1 << k :].any()


def test_msm_vs_naive_and_srs_consistency(orc):
    prm = orc.params_new(8)
    g, gl = prm.g_np(), prm.g_lagrange_np()
    n = 1 << 8
    L = orc.load()
    for i in (0, 1, n - 1):
        assert L.orc_g1a_on_curve(p(np.ascontiguousarray(g[i]))) == 1
        assert L.orc_g1a_on_curve(p(np.ascontiguousarray(gl[i]))) == 1
    s = orc.fill_fr(3, n)
    want = orc.msm_naive(s, g)
    assert np.array_equal(orc.msm(s, g), want)
    assert np.array_equal(orc.msm(s, g, threads=4), want)
    sp = orc.fill_fr_sparse(4, n)
    assert np.array_equal(orc.msm(sp, g), orc.msm_naive(sp, g))
    # KZG consistency: commit(coeffs) over g == commit_lagrange(evals) over g_lagrange
    d = orc.domain(6, 8)
    ev = orc.fft(s, d.fe("omega"))
    assert np.array_equal(orc.msm(ev, gl), want)
    # g[0] = G = (1, 2); g[1] = s*G
    assert orc.fq_to_int(g[0][:4]) == 1 and orc.fq_to_int(g[0][4:]) == 2


def test_chacha20_block_rfc7539(orc):
    """The blinding generator's block function against RFC 7539 section 2.3.2 (key 00..1f, nonce 000000090000004a00000000,
    block counter 1)."""
    key = bytes(range(32))
    nonce = bytes([0, 0, 0, 9, 0, 0, 0, 0x4A, 0, 0, 0, 0])
    words = [int.from_bytes(nonce[4 * i:4 * i + 4], "little") for i in range(3)]
    want = ("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
            "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")
    assert orc.chacha20_block(key, 1, words).hex() == want


def test_rand_fr_is_the_first_keystream_candidate_below_r(orc):
    """rand_fr(key, tag, index): nonce = (tag, index_lo, index_hi), block counter = attempt, two 254-bit candidates per
    block, the first one below r; returned in Montgomery form."""
    r = orc.FR_MODULUS if hasattr(orc, "FR_MODULUS") else 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    key = bytes(range(100, 132))
    seen_second, seen_retry = False, False
    for tag, index in [(1, 0), (6, 12345), (3, (7 << 32) | 9)] + [(2, i) for i in range(40)]:
        want = None
        for attempt in range(8):
            blk = orc.chacha20_block(key, attempt, [tag, index & 0xFFFFFFFF, index >> 32])
            for half in range(2):
                v = int.from_bytes(blk[32 * half:32 * half + 32], "little") & ((1 << 254) - 1)
                if v < r:
                    want = v
                    seen_second |= half == 1
                    seen_retry |= attempt > 0
                    break
            if want is not None:
                break
        got = orc.rand_fr(key, tag, index)
        mont = sum(int(got[i]) << (64 * i) for i in range(4))
        assert mont * pow(1 << 256, -1, r) % r == want
    assert seen_second  # (the rejection path is exercised: 24 % of first candidates are >= r)
