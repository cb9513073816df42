"""Round trips of the on-disk artefacts around the proving path (0g-halo2_amd/formats.py; layouts of
/root/reference/src/io.rs:137-207 as published upstream -- no file from the real CLI exists to compare)."""
import json
import struct

import numpy as np
import pytest

import formats
import wnn_model as wm


def test_srs_roundtrip_and_layout(tmp_path, orc):
    k = 5
    prm = orc.params_new(k, 0xABC)
    g, gl = prm.g_np(), prm.g_lagrange_np()
    g2 = np.frombuffer(bytes(prm.g2), np.uint64)
    s_g2 = np.frombuffer(bytes(prm.s_g2), np.uint64)
    path = str(tmp_path / "kzg.srs")
    formats.write_srs(path, k, g, gl, g2, s_g2)
    raw = open(path, "rb").read()
    assert len(raw) == 4 + 2 * 32 * 64 + 2 * 128 and struct.unpack("<I", raw[:4])[0] == k
    # g[0] = generator (1, 2): Montgomery form of 1 in the first 32 bytes
    one = (1 << 256) % 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    assert int.from_bytes(raw[4:36], "little") == one
    k2, g_r, gl_r, g2_r, sg2_r = formats.read_srs(path)
    assert k2 == k and np.array_equal(g_r, g) and np.array_equal(gl_r, gl)
    assert np.array_equal(g2_r, g2) and np.array_equal(sg2_r, s_g2)
    # the parsed SRS still proves and pairing-verifies (g2 / s_g2 survived the trip)
    open(path, "ab").write(b"\0")
    with pytest.raises(ValueError):
        formats.read_srs(path)


def test_circuit_params_json(tmp_path):
    p = wm.load_checked_in(wm.MNIST_SMALL[1]).get_circuit_params()
    path = str(tmp_path / "params.json")
    formats.write_circuit_params(path, p)
    assert json.load(open(path)) == {"p": 2097143, "l": 20, "n_hashes": 2, "bits_per_hash": 10,
                                     "bits_per_filter": 28, "n_classes": 10}
    assert formats.read_circuit_params(path) == p


def test_proof_with_output_json(tmp_path):
    scores = [9, 6, 13, 10, 17, 10, 9, 26, 11, 16]
    pw = formats.ProofWithOutput(bytes(range(200)), scores)
    path = str(tmp_path / "proof.json")
    pw.write(path, form="hex")
    d = json.load(open(path))
    assert d["proof"][:3] == [0, 1, 2] and d["output"][0] == "0x09" + "00" * 31
    back = formats.ProofWithOutput.read(path)
    assert back.proof == pw.proof and back.output == scores
    # the other candidate encoding: four Montgomery limbs per element (the ABI's zg_fr); read() takes either
    pw.write(path)
    d = json.load(open(path))
    R_ = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    assert sum(l << (64 * j) for j, l in enumerate(d["output"][0])) == 9 * (1 << 256) % R_
    back = formats.ProofWithOutput.read(path)
    assert back.proof == pw.proof and back.output == scores
    m = back.output_mont()
    assert m.shape == (1, 10, 4)
    R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    assert sum(int(m[0, 7, j]) << (64 * j) for j in range(4)) == 26 * (1 << 256) % R
    with pytest.raises(ValueError):
        formats.fr_from_repr_hex("0x" + "ff" * 32)


def test_proving_key_file_roundtrip_and_layout(tmp_path, orc):
    """ProvingKey::write(RawBytes) (io.rs:159-162): written field by field as documented in formats.py, read back, and the
    two families zg_prover_create consumes (fixed values, sigma values) handed on unchanged."""
    from circuits import toy_circuit

    k = 5
    cs, asg, ilen = toy_circuit(k, force_degree=6)
    n, en = 1 << k, 1 << cs.extended_k()
    fixed, sigma = asg.fixed_values(), asg.sigma_values()
    F, P = fixed.shape[0], sigma.shape[0]
    rng = np.random.default_rng(3)
    num_selectors = 3
    sel = rng.integers(0, 2, size=(num_selectors, n)).astype(bool)

    def rnd(*shape):
        return rng.integers(0, 1 << 62, size=shape + (4,), dtype=np.uint64)

    pk = formats.ProvingKeyFile(
        k, fixed_commitments=rng.integers(0, 1 << 62, size=(F, 8), dtype=np.uint64),
        permutation_commitments=rng.integers(0, 1 << 62, size=(P, 8), dtype=np.uint64), selectors=sel,
        l0=rnd(en), l_last=rnd(en), l_active_row=rnd(en), fixed_values=fixed, fixed_polys=rnd(F, n), fixed_cosets=rnd(F, en),
        permutations=sigma, permutation_polys=rnd(P, n), permutation_cosets=rnd(P, en))
    path = str(tmp_path / "pk.bin")
    formats.write_pk(path, pk)
    raw = open(path, "rb").read()
    # header: k and the fixed-commitment count, both u32 big-endian
    assert struct.unpack(">II", raw[:8]) == (k, F)
    sel_at = 8 + 64 * (F + P)
    assert raw[sel_at] == sum(int(sel[0, i]) << i for i in range(8))  # 8 rows per byte, lowest row = bit 0
    l0_at = sel_at + num_selectors * (n // 8)
    assert struct.unpack(">I", raw[l0_at:l0_at + 4])[0] == en
    total = 8 + 64 * (F + P) + num_selectors * n // 8 + 3 * (4 + 32 * en) + 2 * (4 + F * (4 + 32 * n)) + (4 + F * (4 + 32 * en)) \
        + 2 * (4 + P * (4 + 32 * n)) + (4 + P * (4 + 32 * en))
    assert len(raw) == total
    back = formats.read_pk(path, num_selectors, P)
    assert back.k == k
    for f in formats.ProvingKeyFile.FIELDS:
        assert np.array_equal(np.asarray(getattr(back, f)), np.asarray(getattr(pk, f))), f
    # what the backend takes from the file is what keygen produced
    assert np.array_equal(back.fixed_values, fixed) and np.array_equal(back.permutations, sigma)
    with pytest.raises(ValueError):
        formats.read_pk(path, num_selectors + 1, P)
    open(path, "ab").write(b"\0")
    with pytest.raises(ValueError):
        formats.read_pk(path, num_selectors, P)
