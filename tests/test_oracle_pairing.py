"""The oracle's BN254 pairing (oracle/pairing.c) and the pairing-based KZG/GWC verifier built on it.

Pinned by: the G2 generator's curve/order checks, bilinearity and non-degeneracy, a re-computation of
one pairing value with Python integers only (same definition: reduced Tate pairing, Miller loop over r,
exponent (q^12-1)/r), and agreement with the toxic-scalar verifier on good and tampered proofs."""
import ctypes

import numpy as np

from circuits import toy_circuit

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


def p(a):
    return ctypes.c_void_p(a.ctypes.data)


def g1_mul(orc, k):
    L = orc.load()
    g, o, a = np.zeros(12, np.uint64), np.zeros(12, np.uint64), np.zeros(8, np.uint64)
    L.orc_g1_generator(p(g))
    kk = orc.fr_from_int(k)
    L.orc_g1_mul(p(o), p(g), p(kk))
    L.orc_g1_to_affine(p(a), p(o))
    return a


def g2_mul(orc, k):
    L = orc.load()
    g, o = np.zeros(16, np.uint64), np.zeros(16, np.uint64)
    L.orc_g2_generator(p(g))
    kk = orc.fr_from_int(k)
    L.orc_g2a_mul(p(o), p(g), p(kk))
    return o


def pair(orc, a, b):
    o = np.zeros(48, np.uint64)
    orc.load().orc_pairing(p(o), p(a), p(b))
    return o


def test_g2_generator_and_group_law(orc):
    L = orc.load()
    g = g2_mul(orc, 1)
    assert L.orc_g2a_on_curve(p(g)) == 1
    # EIP-197 / py_ecc generator of the order-r subgroup of the twist y^2 = x^3 + 3/(9+i)
    assert orc.fq_to_int(g[0:4]) == 10857046999023057135944570762232829481370756359578518086990519993285655852781
    assert orc.fq_to_int(g[4:8]) == 11559732032986387107991004021392285783925812861821192530917403151452391805634
    assert orc.fq_to_int(g[8:12]) == 8495653923123431417604973247489272438418190587263600148770280649306958101930
    assert orc.fq_to_int(g[12:16]) == 4082367875863433681332203403145435568316851327593401208105741076214120093531
    o = np.zeros(16, np.uint64)
    m1, g5, g6, g7, g12 = (g2_mul(orc, v) for v in (R - 1, 5, 6, 7, 12))  # (named: ctypes borrows the buffers)
    L.orc_g2a_add(p(o), p(m1), p(g))
    assert L.orc_g2a_is_identity(p(o)) == 1
    L.orc_g2a_add(p(o), p(g5), p(g7))
    assert L.orc_g2a_eq(p(o), p(g12)) == 1 and L.orc_g2a_on_curve(p(o)) == 1
    L.orc_g2a_add(p(o), p(g6), p(g6))  # doubling branch
    assert L.orc_g2a_eq(p(o), p(g12)) == 1


def test_bilinear_nondegenerate(orc):
    L = orc.load()
    g1, g2 = g1_mul(orc, 1), g2_mul(orc, 1)
    e = pair(orc, g1, g2)
    assert L.orc_fq12_is_one(p(e)) == 0
    a, b = 0x1234567890ABCDEF, R - 12345
    w, kab, km1 = np.zeros(48, np.uint64), orc.fr_from_int(a * b % R), orc.fr_from_int(R - 1)
    L.orc_fq12_pow_fr(p(w), p(e), p(kab))
    eab = pair(orc, g1_mul(orc, a), g2_mul(orc, b))
    assert L.orc_fq12_eq(p(eab), p(w)) == 1
    # e^r = 1, e(O, Q) = e(P, O) = 1
    L.orc_fq12_pow_fr(p(w), p(e), p(km1))
    m = np.zeros(48, np.uint64)
    L.orc_fq12_mul(p(m), p(w), p(e))
    assert L.orc_fq12_is_one(p(m)) == 1
    e1, e2 = pair(orc, np.zeros(8, np.uint64), g2), pair(orc, g1, np.zeros(16, np.uint64))
    assert L.orc_fq12_is_one(p(e1)) == 1 and L.orc_fq12_is_one(p(e2)) == 1
    # product check: e(aG, [1]_2) e(-G, [a]_2) = 1, and fails for a wrong exponent
    ps = np.stack([g1_mul(orc, a), g1_mul(orc, R - 1)])
    good, bad = np.stack([g2, g2_mul(orc, a)]), np.stack([g2, g2_mul(orc, a + 1)])
    assert L.orc_pairing_check(p(ps), p(good), ctypes.c_size_t(2)) == 1
    assert L.orc_pairing_check(p(ps), p(bad), ctypes.c_size_t(2)) == 0


# ---- the same pairing with Python integers only ----
def f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def f12mul(a, b):
    acc = [(0, 0)] * 11
    for i in range(6):
        if a[i] == (0, 0):
            continue
        for j in range(6):
            t = f2mul(a[i], b[j])
            acc[i + j] = ((acc[i + j][0] + t[0]) % Q, (acc[i + j][1] + t[1]) % Q)
    for d in range(10, 5, -1):
        t = f2mul(acc[d], (9, 1))
        acc[d - 6] = ((acc[d - 6][0] + t[0]) % Q, (acc[d - 6][1] + t[1]) % Q)
    return acc[:6]


def tate_python(P, Qp):
    (xp, yp), (xq, yq) = P, Qp
    one = [(1, 0)] + [(0, 0)] * 5
    f, xt, yt = one, xp, yp

    def line(lam):
        return [((lam * xt - yt) % Q, 0), (0, 0), ((-lam * xq[0]) % Q, (-lam * xq[1]) % Q), yq, (0, 0), (0, 0)]

    bits = bin(R)[3:]
    inf = False
    for bit in bits:
        f = f12mul(f, f)
        lam = 3 * xt * xt * pow(2 * yt, -1, Q) % Q
        f = f12mul(f, line(lam))
        x3 = (lam * lam - 2 * xt) % Q
        xt, yt = x3, (lam * (xt - x3) - yt) % Q
        if bit == "1":
            if xt == xp:
                assert (yt + yp) % Q == 0
                inf = True  # vertical line, last step
                continue
            lam = (yp - yt) * pow(xp - xt, -1, Q) % Q
            f = f12mul(f, line(lam))
            x3 = (lam * lam - xt - xp) % Q
            xt, yt = x3, (lam * (xt - x3) - yt) % Q
    assert inf
    e = (Q**12 - 1) // R
    acc = one
    for bit in bin(e)[2:]:
        acc = f12mul(acc, acc)
        if bit == "1":
            acc = f12mul(acc, f)
    return acc


def test_pairing_value_matches_python_integers(orc):
    a = g1_mul(orc, 7)
    b = g2_mul(orc, 11)
    P = (orc.fq_to_int(a[:4]), orc.fq_to_int(a[4:]))
    Q2 = ((orc.fq_to_int(b[0:4]), orc.fq_to_int(b[4:8])), (orc.fq_to_int(b[8:12]), orc.fq_to_int(b[12:16])))
    want = tate_python(P, Q2)
    got = pair(orc, a, b)
    got_int = [(orc.fq_to_int(got[8 * i:8 * i + 4]), orc.fq_to_int(got[8 * i + 4:8 * i + 8])) for i in range(6)]
    assert got_int == [tuple(x) for x in want]


def test_pairing_verifier_accepts_good_rejects_tampered(orc):
    k = 6
    cs, asg, ilen = toy_circuit(k)
    params = orc.params_new(k, 0xABCDEF)
    L = orc.load()
    assert L.orc_g2a_on_curve(ctypes.byref(params, type(params).s_g2.offset)) == 1
    pk = orc.ProvingKey(cs.to_c(), asg.fixed_values(), asg.sigma_values(), params, orc.fr_from_int(0x1234567))
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    st, proof, _ = orc.create_proof(pk, adv, inst, 3)
    assert st == 0
    assert orc.verify_proof(pk, inst, proof) == 1
    assert orc.verify_proof_pairing(pk, inst, proof) == 1
    # any flipped evaluation / commitment bit must fail under both verifiers (or be rejected as malformed)
    for pos in (5, len(proof) // 2, len(proof) - 70, len(proof) - 1):
        bad = bytearray(proof)
        bad[pos] ^= 1
        assert orc.verify_proof_pairing(pk, inst, bytes(bad)) != 1
        assert orc.verify_proof(pk, inst, bytes(bad)) != 1
    # wrong public input
    inst2 = inst.copy()
    inst2[0, 0] = orc.fr_from_int(orc.fr_to_int(inst[0, 0]) + 1)
    assert orc.verify_proof_pairing(pk, inst2, proof) != 1
