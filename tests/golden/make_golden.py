#!/usr/bin/env python3
"""Writes tests/golden/vectors.json -- run from the repo root:  python tests/golden/make_golden.py

Three kinds of vectors (the reference holds no byte-level KAT for this path, so these are
what pins it):
  * "bigint": inputs + expected outputs computed HERE with Python integers only (affine chord-tangent
    group law, O(n^2) DFT) -- independent of oracle/*.c and of the HIP code; both must reproduce them.
  * "reference": data the reference's own tests hold (tests/integration_test.rs snapshot predictions,
    the Keccak-256 empty-string digest its EVM transcript relies on).
  * "oracle": outputs of oracle/*.c on seeded inputs (proof bytes, large-size digests), frozen so that
    a later change to either side shows up as a diff against a committed file.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
sys.path.insert(0, os.path.join(ROOT, "harness"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


# ---------------------------------------------------------------- big-integer BN254 G1 (y^2 = x^3 + 3)
def ec_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return x3, (lam * (x1 - x3) - y1) % Q


def ec_mul(p, k):
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, p)
        p = ec_add(p, p)
        k >>= 1
    return acc


def splitmix_scalars(seed, n, sparse):
    """Deterministic scalars: the advice-like mix of SURVEY.md 8d (zeros, bits, bytes, full width)."""
    out, x = [], seed & (2**64 - 1)

    def nxt():
        nonlocal x
        x = (x + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)

    for _ in range(n):
        v = (nxt() | (nxt() << 64) | (nxt() << 128) | (nxt() << 192)) % R
        if sparse:
            sel = nxt() % 100
            v = 0 if sel < 60 else (v & 1) if sel < 80 else (v & 0xFF) if sel < 92 else (R - 1 - (v & 3)) if sel < 95 else v
        out.append(v)
    return out


def hx(v, nbytes=32):
    return format(v, "0%dx" % (2 * nbytes))


def bigint_vectors():
    vec = []
    g = (1, 2)
    for name, n, seed, sparse, s_tau in (("msm_dense_33", 33, 11, False, 0x1234567), ("msm_sparse_64", 64, 12, True, 0x5EED),
                                         ("msm_single", 1, 13, False, 0x77), ("msm_repeated_base", 16, 14, False, 1)):
        # bases tau^i * G (tau = 1 gives 16 copies of G: doubling / cancellation inside one bucket)
        bases, t = [], 1
        for _ in range(n):
            bases.append(ec_mul(g, t))
            t = t * s_tau % R
        scal = splitmix_scalars(seed, n, sparse)
        if name == "msm_repeated_base":
            scal[3] = (R - scal[2]) % R  # s*G + (-s)*G inside the sum
        acc = None
        for s, b in zip(scal, bases):
            if s:
                acc = ec_add(acc, ec_mul(b, s))
        vec.append({"name": name, "kind": "msm", "bases": [[hx(b[0]), hx(b[1])] for b in bases],
                    "scalars": [hx(s) for s in scal],
                    "result": None if acc is None else [hx(acc[0]), hx(acc[1])]})
    # all-zero scalars and a sum that cancels to the identity
    vec.append({"name": "msm_zero_scalars", "kind": "msm", "bases": [[hx(1), hx(2)]] * 8, "scalars": [hx(0)] * 8,
                "result": None})
    vec.append({"name": "msm_cancels", "kind": "msm", "bases": [[hx(1), hx(2)]] * 2, "scalars": [hx(5), hx(R - 5)],
                "result": None})
    for log_n, seed in ((3, 21), (6, 22)):
        n = 1 << log_n
        om = pow(pow(7, (R - 1) >> 28, R), 1 << (28 - log_n), R)
        a = splitmix_scalars(seed, n, False)
        f = [sum(a[j] * pow(om, j * i, R) for j in range(n)) % R for i in range(n)]
        zeta = pow(7, (R - 1) // 3, R)
        ext_k = log_n + 3
        eom = pow(pow(7, (R - 1) >> 28, R), 1 << (28 - ext_k), R)
        ext = [sum(c * pow(zeta * pow(eom, i, R) % R, j, R) for j, c in enumerate(a)) % R for i in range(1 << ext_k)]
        vec.append({"name": f"ntt_{n}", "kind": "ntt", "log_n": log_n, "omega": hx(om), "input": [hx(v) for v in a],
                    "output": [hx(v) for v in f],
                    "coset_ext_k": ext_k, "coset_output_sha256": hashlib.sha256("".join(hx(v) for v in ext).encode()).hexdigest()})
    return vec


def oracle_vectors():
    import orc
    import wnn_shape
    from circuits import toy_circuit

    out = []

    def proof_case(name, cs, asg, ilen, k, srs_seed, vk, seed):
        img = cs.to_c()
        params = orc.params_new(k, srs_seed)
        pk = orc.ProvingKey(img, asg.fixed_values(), asg.sigma_values(), params, orc.fr_from_int(vk))
        st, proof, _ = orc.create_proof(pk, asg.advice_values(), asg.instance_values(ilen), seed)
        assert st == 0 and orc.verify_proof(pk, asg.instance_values(ilen), proof) == 1
        return {"name": name, "kind": "proof", "k": k, "srs_seed": srs_seed, "vk_repr": vk, "blinding_seed": seed,
                "proof_len": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(), "proof_hex": proof.hex()}

    cs, asg, ilen = toy_circuit(5)
    out.append(proof_case("toy_k5", cs, asg, ilen, 5, 0xABCDEF, 0x1234567, 1))
    cs, asg, ilen = toy_circuit(8, force_degree=6)
    out.append(proof_case("toy_k8_degree6", cs, asg, ilen, 8, 0xABCDEF, 0x1234567, 2))
    cs, asg, ilen = wnn_shape.build("tiny", k=12, seed=1)
    out.append(proof_case("wnn_shape_k12", cs, asg, ilen, 12, 0x5EED, 0xC0FFEE, 7))
    # zero_g's real WnnCircuit for model_28input_256entry_1hash_1bpi on example_image_7 (k = 14): digest only
    import wnn_circuit
    import wnn_model

    k, name = wnn_model.MNIST_TINY
    cs, asg, ilen, scores = wnn_circuit.build(wnn_model.load_checked_in(name), wnn_model.load_test_image(), k)
    orc.load().orc_set_threads(8)
    case = proof_case("wnn_real_tiny_k14", cs, asg, ilen, k, 0x5EED, 0xC0FFEE, 7)
    del case["proof_hex"]
    case["class_scores"] = scores
    out.append(case)
    # large-size digests: MSM 2^14 over the seeded SRS, iNTT 2^14, coset NTT 2^14 -> 2^17
    k = 14
    prm = orc.params_new(k)
    s = orc.fill_fr_sparse(11, 1 << k)
    r = orc.msm(s, prm.g_lagrange_np(), threads=8)
    out.append({"name": "msm_2p14_sparse_seed11", "kind": "digest", "sha256": hashlib.sha256(r.tobytes()).hexdigest()})
    s = orc.fill_fr(12, 1 << k)
    r = orc.msm(s, prm.g_np(), threads=8)
    out.append({"name": "msm_2p14_dense_seed12", "kind": "digest", "sha256": hashlib.sha256(r.tobytes()).hexdigest()})
    d = orc.domain(6, k)
    a = orc.fill_fr(13, 1 << k)
    coeff = orc.lagrange_to_coeff(d, a)
    out.append({"name": "intt_2p14_seed13", "kind": "digest", "sha256": hashlib.sha256(coeff.tobytes()).hexdigest()})
    ext = orc.coeff_to_extended(d, coeff)
    out.append({"name": "coset_2p17_seed13", "kind": "digest", "sha256": hashlib.sha256(ext.tobytes()).hexdigest()})
    return out


REFERENCE = {
    # /root/reference/tests/integration_test.rs:19,36,53,70 -- Wnn::predict on benches/example_image_7.png
    "predictions": {
        "model_28input_256entry_1hash_1bpi": [9, 6, 13, 10, 17, 10, 9, 26, 11, 16],
        "model_28input_1024entry_2hash_2bpi": [17, 13, 25, 27, 29, 21, 15, 55, 27, 32],
        "model_28input_2048entry_2hash_3bpi": [29, 21, 40, 47, 45, 41, 28, 82, 35, 66],
        "model_49input_8192entry_4hash_6bpi": [16, 10, 22, 22, 29, 25, 9, 91, 21, 51],
    },
    # Keccak-256 (not SHA3-256) of the empty string and of "abc": the hash EvmTranscript is built on
    "keccak256": {"": "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470",
                  "616263": "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"},
}

if __name__ == "__main__":
    doc = {"bigint": bigint_vectors(), "reference": REFERENCE, "oracle": oracle_vectors()}
    path = os.path.join(ROOT, "tests", "golden", "vectors.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")
