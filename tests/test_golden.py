"""Committed golden vectors (tests/golden/vectors.json, made by tests/golden/make_golden.py):
  * "bigint" vectors were computed with Python integers only -- the oracle (CPU, here) and the HIP path
    (GPU, through the C ABI) must both reproduce them bit for bit;
  * "reference" holds the data the reference's own tests carry for this path;
  * "oracle" vectors freeze proof bytes and large-size digests.
"""
import hashlib
import json
import os

import numpy as np
import pytest

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
MONT = 1 << 256
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "vectors.json")) as f:
        return json.load(f)


def limbs(x):
    return [(x >> (64 * i)) & (2**64 - 1) for i in range(4)]


def fr_mont(vals):
    return np.array([limbs(int(v, 16) * MONT % R) for v in vals], dtype=np.uint64).reshape(-1, 4)


def fr_raw_hex(arr, to_int):
    return [format(to_int(a), "064x") for a in arr]


def bases_mont(pts):
    return np.array([limbs(int(x, 16) * MONT % Q) + limbs(int(y, 16) * MONT % Q) for x, y in pts],
                    dtype=np.uint64).reshape(-1, 8)


def point_expect(res):
    """normalised Jacobian limbs of a golden affine result (None = identity (0, 1, 0))"""
    if res is None:
        return np.array(limbs(0) + limbs(MONT % Q) + limbs(0), dtype=np.uint64)
    return np.array(limbs(int(res[0], 16) * MONT % Q) + limbs(int(res[1], 16) * MONT % Q) + limbs(MONT % Q),
                    dtype=np.uint64)


# ------------------------------------------------------------------ CPU: the oracle against the vectors
def test_oracle_msm_matches_bigint_vectors(orc, golden):
    cases = [v for v in golden["bigint"] if v["kind"] == "msm"]
    assert len(cases) >= 6
    for v in cases:
        got = orc.msm(fr_mont(v["scalars"]), bases_mont(v["bases"]))
        assert np.array_equal(got, point_expect(v["result"])), v["name"]
        assert np.array_equal(orc.msm_naive(fr_mont(v["scalars"]), bases_mont(v["bases"])), point_expect(v["result"]))


def test_oracle_ntt_matches_bigint_vectors(orc, golden):
    for v in (v for v in golden["bigint"] if v["kind"] == "ntt"):
        a = fr_mont(v["input"])
        d = orc.domain(6, v["log_n"])
        assert format(orc.fr_to_int(d.fe("omega")), "064x") == v["omega"]
        f = orc.fft(a, d.fe("omega"))
        assert fr_raw_hex(f, orc.fr_to_int) == v["output"], v["name"]
        assert d.extended_k == v["coset_ext_k"]
        ext = orc.coeff_to_extended(d, a)
        assert hashlib.sha256("".join(fr_raw_hex(ext, orc.fr_to_int)).encode()).hexdigest() == v["coset_output_sha256"]


def test_oracle_keccak_reference_vectors(orc, golden):
    for msg, dig in golden["reference"]["keccak256"].items():
        assert orc.keccak256(bytes.fromhex(msg)).hex() == dig


def _proof_case(v):
    import wnn_shape
    from circuits import toy_circuit

    if v["name"] == "toy_k5":
        return toy_circuit(5)
    if v["name"] == "toy_k8_degree6":
        return toy_circuit(8, force_degree=6)
    if v["name"] == "wnn_real_tiny_k14":
        import wnn_circuit
        import wnn_model

        k, name = wnn_model.MNIST_TINY
        cs, asg, ilen, scores = wnn_circuit.build(wnn_model.load_checked_in(name), wnn_model.load_test_image(), k)
        assert scores == v["class_scores"]
        return cs, asg, ilen
    assert v["name"] == "wnn_shape_k12"
    return wnn_shape.build("tiny", k=12, seed=1)


def test_oracle_reproduces_frozen_proofs(orc, golden):
    for v in (v for v in golden["oracle"] if v["kind"] == "proof"):
        cs, asg, ilen = _proof_case(v)
        params = orc.params_new(v["k"], v["srs_seed"])
        pk = orc.ProvingKey(cs.to_c(), asg.fixed_values(), asg.sigma_values(), params, orc.fr_from_int(v["vk_repr"]))
        orc.load().orc_set_threads(8)
        st, proof, _ = orc.create_proof(pk, asg.advice_values(), asg.instance_values(ilen), v["blinding_seed"])
        assert st == 0 and len(proof) == v["proof_len"]
        assert hashlib.sha256(proof).hexdigest() == v["proof_sha256"], v["name"]
        if "proof_hex" in v:
            assert proof.hex() == v["proof_hex"], v["name"]
        assert orc.verify_proof_pairing(pk, asg.instance_values(ilen), proof) == 1


# ------------------------------------------------------------------ GPU: the HIP path against the vectors
@pytest.mark.gpu
def test_hip_msm_matches_bigint_vectors(ctx, zg, golden):
    for v in (v for v in golden["bigint"] if v["kind"] == "msm"):
        for c in (0, 5):
            bases = ctx.register_bases(bases_mont(v["bases"]), c)
            got = ctx.msm(bases, fr_mont(v["scalars"]))
            bases.free()
            assert np.array_equal(got, point_expect(v["result"])), (v["name"], c)


@pytest.mark.gpu
def test_hip_ntt_matches_bigint_vectors(ctx, zg, golden):
    for v in (v for v in golden["bigint"] if v["kind"] == "ntt"):
        a = fr_mont(v["input"])
        omega = fr_mont([v["omega"]])[0]
        f = ctx.ntt(a, omega)
        assert fr_raw_hex(f, zg.fr_to_int) == v["output"], v["name"]
        ext = ctx.coeff_to_extended(a, v["log_n"], v["coset_ext_k"])
        assert hashlib.sha256("".join(fr_raw_hex(ext, zg.fr_to_int)).encode()).hexdigest() == v["coset_output_sha256"]


@pytest.mark.gpu
def test_hip_keccak_reference_vectors(zg, golden):
    for msg, dig in golden["reference"]["keccak256"].items():
        assert zg.keccak256(bytes.fromhex(msg)).hex() == dig


@pytest.mark.gpu
def test_hip_reproduces_frozen_proofs(ctx, zg, orc, golden):
    for v in (v for v in golden["oracle"] if v["kind"] == "proof"):
        cs, asg, ilen = _proof_case(v)
        params = orc.params_new(v["k"], v["srs_seed"])  # SRS generation only; the proof comes from the GPU
        prover = zg.Prover(ctx, cs.to_c(), asg.fixed_values(), asg.sigma_values(), params.g_np(),
                           params.g_lagrange_np(), orc.fr_from_int(v["vk_repr"]))
        proof = prover.prove(asg.advice_values(), asg.instance_values(ilen), v["blinding_seed"])
        prover.close()
        assert hashlib.sha256(proof).hexdigest() == v["proof_sha256"], v["name"]
        if "proof_hex" in v:
            assert proof.hex() == v["proof_hex"], v["name"]


@pytest.mark.gpu
def test_hip_full_size_digests(ctx, zg, orc, golden):
    dig = {v["name"]: v["sha256"] for v in golden["oracle"] if v["kind"] == "digest"}
    k = 14
    prm = orc.params_new(k)  # inputs only
    bl = ctx.register_bases(prm.g_lagrange_np())
    assert hashlib.sha256(ctx.msm(bl, orc.fill_fr_sparse(11, 1 << k)).tobytes()).hexdigest() == dig["msm_2p14_sparse_seed11"]
    bl.free()
    bg = ctx.register_bases(prm.g_np())
    assert hashlib.sha256(ctx.msm(bg, orc.fill_fr(12, 1 << k)).tobytes()).hexdigest() == dig["msm_2p14_dense_seed12"]
    bg.free()
    d = orc.domain(6, k)
    coeff = ctx.ntt(orc.fill_fr(13, 1 << k), d.fe("omega_inv"), d.fe("ifft_divisor"))
    assert hashlib.sha256(coeff.tobytes()).hexdigest() == dig["intt_2p14_seed13"]
    ext = ctx.coeff_to_extended(coeff, k, d.extended_k)
    assert hashlib.sha256(ext.tobytes()).hexdigest() == dig["coset_2p17_seed13"]
