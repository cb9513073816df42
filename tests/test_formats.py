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
    pw.write(path)
    d = json.load(open(path))
    assert d["proof"][:3] == [0, 1, 2] and d["output"][0] == "0x09" + "00" * 31
    back = formats.ProofWithOutput.read(path)
    assert back.proof == pw.proof and back.output == scores
    m = back.output_mont()
    assert m.shape == (1, 10, 4)
    R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    assert sum(int(m[0, 7, j]) << (64 * j) for j in range(4)) == 26 * (1 << 256) % R
    with pytest.raises(ValueError):
        formats.fr_from_repr_hex("0x" + "ff" * 32)
