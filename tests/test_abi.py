"""CPU-side checks of the drop-in boundary: libzg_halo2.so loads without a GPU, exports every symbol
include/zg_halo2.h declares, and refuses (loudly, no CPU fallback) to create a context with no device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "zg_halo2.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(zg_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_functions():
    names = header_functions()
    assert "zg_msm" in names and "zg_ntt" in names and "zg_ctx_create" in names
    assert len(names) >= 20


def test_library_exports_every_declared_symbol(zg):
    lib = zg.load()
    missing = [n for n in header_functions() if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(zg.ABI_SYMBOLS) == header_functions()
    assert b"gfx950" in lib.zg_version()


def test_no_cpu_fallback_without_device(zg):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(zg.ZgError) as e:
        zg.Ctx(0)
    assert e.value.status == -2  # ZG_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_link_or_import_the_oracle():
    import subprocess

    lib = os.path.join(ROOT, "0g-halo2_amd", "libzg_halo2.so")
    out = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
    assert "zg_oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "0g-halo2_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "orc_" not in src and "import orc" not in src and "zg_oracle" not in src, f


def test_domain_omega_host_helper(zg):
    R = zg.FR_MODULUS
    for log_n in (1, 14, 17, 20, 28):
        om, omi = zg.domain_omega(log_n)
        w, wi = zg.fr_to_int(om), zg.fr_to_int(omi)
        assert w == pow(pow(7, (R - 1) >> 28, R), 1 << (28 - log_n), R)
        assert w * wi % R == 1


def test_g1_sum_host_helper(zg, orc):
    import numpy as np

    prm = orc.params_new(4)
    g = prm.g_np()
    jac = np.zeros((3, 12), np.uint64)
    L = orc.load()
    for i in range(3):
        L.orc_g1_from_affine(ctypes.c_void_p(jac[i].ctypes.data), ctypes.c_void_p(np.ascontiguousarray(g[i]).ctypes.data))
    want = orc.normalise(orc.g1_add(orc.g1_add(jac[0], jac[1]), jac[2]))
    assert np.array_equal(zg.g1_sum(jac), want)
    ident = np.zeros((1, 12), np.uint64)
    got = zg.g1_sum(ident)
    assert not got[:4].any() and not got[8:].any()  # (0, 1, 0)
    assert np.array_equal(zg.g1_sum(jac[:0]), got)


def test_product_keccak_kat(zg):
    """EvmTranscript's hash (host code of the product library; no GPU needed)."""
    assert zg.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert zg.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    import orc

    for n in (1, 31, 32, 33, 135, 136, 137, 272, 1000):
        data = bytes((7 * i + n) & 0xFF for i in range(n))
        assert zg.keccak256(data) == orc.keccak256(data), n
