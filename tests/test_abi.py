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


# ---- the three descriptions of the boundary's structs must agree: C (the header, through gcc -std=c99), Python
# (ctypes Structures of the harness / binding) and Rust (#[repr(C)] structs of shim/.../zg_sys.rs, layout computed here)
def _c_layout(tmp_path):
    import json
    import subprocess

    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi", "layout.c"), "-o", exe], check=True)
    return json.loads(subprocess.run([exe], check=True, capture_output=True, text=True).stdout)


def test_header_is_c99_and_layouts_match_ctypes(zg, tmp_path):
    import circuit as hc

    c = _c_layout(tmp_path)
    pairs = {"zg_query": hc._Query, "zg_monomial": hc._Monomial, "zg_poly": hc._Poly, "zg_lookup": hc._Lookup,
             "zg_circuit": hc.CCircuit, "zg_kernel_stat": zg.KernelStat}
    for name, struct in pairs.items():
        assert ctypes.sizeof(struct) == c[name]["size"], name
        for field, off in c[name]["fields"].items():
            assert getattr(struct, field).offset == off, (name, field)
        assert [f[0] for f in struct._fields_] == list(c[name]["fields"]), name
    # zg_witness_op crosses as a uint64[n, 4] numpy array
    assert c["zg_witness_op"] == {"size": 32, "fields": {"op": 0, "a": 8, "b": 16, "imm": 24}}


def _rust_structs():
    src = open(os.path.join(ROOT, "shim", "halo2_proofs-zg", "src", "zg_sys.rs")).read()
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (\w+): usize = (\d+);", src)}
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[derive\([^)]*\)\])?\s*pub struct (\w+)\s*\{(.*?)\}", src, flags=re.S):
        fields = re.findall(r"pub (\w+): ([^,}]+?)\s*(?:,|$)", m.group(2).strip() + ",")
        out[m.group(1)] = fields
    return out, consts


def _rust_layout(structs, consts, name):
    prim = {"u8": (1, 1), "c_char": (1, 1), "u32": (4, 4), "i32": (4, 4), "u64": (8, 8), "f64": (8, 8), "Fr": (32, 8), "Fq": (32, 8)}

    def size_align(t):
        t = t.strip()
        if t.startswith("*"):
            return 8, 8
        m = re.match(r"\[(.+); (\w+)\]", t)
        if m:
            s, a = size_align(m.group(1))
            n = int(m.group(2)) if m.group(2).isdigit() else consts[m.group(2)]
            return s * n, a
        if t in prim:
            return prim[t]
        return _rust_layout(structs, consts, t)[:2]

    off, align, offsets = 0, 1, {}
    for fname, ftype in structs[name]:
        s, a = size_align(ftype)
        off = (off + a - 1) // a * a
        offsets[fname] = off
        off += s
        align = max(align, a)
    return (off + align - 1) // align * align, align, offsets


def test_rust_repr_c_structs_match_the_header(tmp_path):
    c = _c_layout(tmp_path)
    structs, consts = _rust_structs()
    for name in ("zg_query", "zg_monomial", "zg_poly", "zg_lookup", "zg_circuit", "zg_witness_op", "zg_kernel_stat"):
        size, _, offsets = _rust_layout(structs, consts, name)
        assert size == c[name]["size"], name
        assert offsets == c[name]["fields"], name  # (dict equality + insertion order of both = same field order)
        assert list(offsets) == list(c[name]["fields"]), name


def test_rust_extern_block_is_generated_from_the_header():
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_zg_sys.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    src = open(os.path.join(ROOT, "shim", "halo2_proofs-zg", "src", "zg_sys.rs")).read()
    declared = sorted(set(re.findall(r"pub fn (zg_[a-z0-9_]+)\(", src)))
    assert declared == header_functions()


def test_plain_c_driver_builds_against_the_library(tmp_path):
    """tests/abi/c_driver.c (C99, -pedantic) links against libzg_halo2.so with nothing but the header; without a GPU it
    stops at zg_ctx_create with the library's own no-device error (the run itself is tests/test_gpu_abi_c.py)."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi", "gen_c_driver_data.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = str(tmp_path / "c_driver")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I",
                    os.path.join(ROOT, "tests", "abi"), os.path.join(ROOT, "tests", "abi", "c_driver.c"), "-L",
                    os.path.join(ROOT, "0g-halo2_amd"), "-lzg_halo2", "-Wl,-rpath," + os.path.join(ROOT, "0g-halo2_amd"), "-o", exe],
                   check=True)
    import torch

    if not torch.cuda.is_available():
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


def test_field_inv_returns_zero_when_its_budget_runs_out_host_path():
    """Field::inv on a non-canonical multiple of the modulus must terminate AND return 0, not a would-be inverse
    (tests/abi/field_probe.hip, built by the package Makefile; the device path runs in tests/test_gpu_msm.py)."""
    import subprocess

    exe = os.path.join(ROOT, "tests", "abi", "field_probe")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe, "host"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
