"""ctypes binding of oracle/libzg_oracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Arrays use the same numpy conventions as the product binding: Fr/Fq = uint64[..., 4] Montgomery
limbs, affine = uint64[..., 8], Jacobian = uint64[..., 12].
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_int, c_int32, c_size_t, c_uint32, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzg_oracle.so")
_lib = None


def build() -> None:
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        # OpenMP would start one thread per core of the MACHINE; a box that grants this process a share of them
        # (cgroup / affinity) then thrashes -- 35 s instead of 0.3 s for a k = 9 proof on the GPU box.  Default to the
        # cores this process may use, 16 at most; callers may raise or lower it (orc_set_threads).
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        _lib.orc_set_threads(max(1, min(cores, 16)))
    return _lib


def _p(a: np.ndarray) -> c_void_p:
    return c_void_p(a.ctypes.data)


def _c(a, last) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.shape[-1] == last, a.shape
    return a


class Domain(ctypes.Structure):
    _fields_ = (
        [("k", c_uint32), ("extended_k", c_uint32), ("n", c_uint64), ("extended_n", c_uint64),
         ("quotient_poly_degree", c_uint32)]
        + [(name, c_uint64 * 4) for name in (
            "omega", "omega_inv", "extended_omega", "extended_omega_inv", "g_coset", "g_coset_inv",
            "ifft_divisor", "extended_ifft_divisor", "barycentric_weight")]
        + [("t_evaluations", c_void_p), ("t_len", c_size_t)]
    )

    def fe(self, name: str) -> np.ndarray:
        return np.array(list(getattr(self, name)), dtype=np.uint64)


def domain(j: int, k: int) -> Domain:
    d = Domain()
    load().orc_domain_new(ctypes.byref(d), c_uint32(j), c_uint32(k))
    return d


class Params(ctypes.Structure):
    _fields_ = [("k", c_uint32), ("n", c_uint64), ("g", c_void_p), ("g_lagrange", c_void_p),
                ("s", c_uint64 * 4), ("g2", c_uint64 * 16), ("s_g2", c_uint64 * 16)]

    def g_np(self) -> np.ndarray:
        n = int(self.n)
        return np.ctypeslib.as_array(ctypes.cast(self.g, ctypes.POINTER(c_uint64)), shape=(n, 8)).copy()

    def g_lagrange_np(self) -> np.ndarray:
        n = int(self.n)
        return np.ctypeslib.as_array(ctypes.cast(self.g_lagrange, ctypes.POINTER(c_uint64)),
                                     shape=(n, 8)).copy()


def params_new(k: int, seed: int = 0x5EED) -> Params:
    """ParamsKZG::new(k) with the toxic scalar drawn from SplitMix64(seed)."""
    s = fill_fr(seed, 1)[0]
    p = Params()
    load().orc_params_new(ctypes.byref(p), c_uint32(k), _p(s))
    return p


def params_from_scalar(k: int, s: np.ndarray) -> Params:
    """ParamsKZG::new(k) for a given toxic scalar (Montgomery limbs)."""
    p = Params()
    load().orc_params_new(ctypes.byref(p), c_uint32(k), _p(np.ascontiguousarray(s, dtype=np.uint64)))
    return p


def fill_fr(seed: int, n: int) -> np.ndarray:
    out = np.zeros((n, 4), np.uint64)
    load().orc_fill_fr(c_uint64(seed), _p(out), c_size_t(n))
    return out


def fill_fr_sparse(seed: int, n: int) -> np.ndarray:
    out = np.zeros((n, 4), np.uint64)
    load().orc_fill_fr_sparse(c_uint64(seed), _p(out), c_size_t(n))
    return out


def fr_from_int(x: int) -> np.ndarray:
    raw = np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    out = np.zeros(4, np.uint64)
    load().orc_fr_from_raw(_p(out), _p(raw))
    return out


def fr_to_int(a) -> int:
    a = _c(a, 4)
    raw = np.zeros(4, np.uint64)
    load().orc_fr_to_raw(_p(raw), _p(a))
    return sum(int(raw[i]) << (64 * i) for i in range(4))


def fq_to_int(a) -> int:
    a = _c(a, 4)
    raw = np.zeros(4, np.uint64)
    load().orc_fq_to_raw(_p(raw), _p(a))
    return sum(int(raw[i]) << (64 * i) for i in range(4))


def fr_inv(a) -> np.ndarray:
    out = np.zeros(4, np.uint64)
    load().orc_fr_inv(_p(out), _p(_c(a, 4)))
    return out


def msm(scalars: np.ndarray, bases: np.ndarray, threads: int = 1) -> np.ndarray:
    """halo2 best_multiexp restated; returns the NORMALISED Jacobian point (z = 1 / identity)."""
    scalars = _c(scalars, 4).reshape(-1, 4)
    bases = _c(bases, 8).reshape(-1, 8)
    n = scalars.shape[0]
    assert bases.shape[0] >= n
    r = np.zeros(12, np.uint64)
    if threads <= 1:
        load().orc_msm(_p(r), _p(scalars), _p(bases), c_size_t(n))
    else:
        load().orc_msm_mt(_p(r), _p(scalars), _p(bases), c_size_t(n), c_int(threads))
    return normalise(r)


def msm_naive(scalars: np.ndarray, bases: np.ndarray) -> np.ndarray:
    scalars = _c(scalars, 4).reshape(-1, 4)
    bases = _c(bases, 8).reshape(-1, 8)
    r = np.zeros(12, np.uint64)
    load().orc_msm_naive(_p(r), _p(scalars), _p(bases), c_size_t(scalars.shape[0]))
    return normalise(r)


def normalise(jac: np.ndarray) -> np.ndarray:
    """Jacobian -> (x, y, 1) or (0, 1, 0): the canonical form the product ABI returns."""
    jac = _c(jac, 12)
    L = load()
    aff = np.zeros(8, np.uint64)
    L.orc_g1_to_affine(_p(aff), _p(jac))
    out = np.zeros(12, np.uint64)
    L.orc_g1_from_affine(_p(out), _p(aff))
    return out


def g1_add(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    out = np.zeros(12, np.uint64)
    load().orc_g1_add(_p(out), _p(_c(p, 12)), _p(_c(q, 12)))
    return out


def fft(a: np.ndarray, omega: np.ndarray) -> np.ndarray:
    a = _c(a, 4).reshape(-1, 4).copy()
    n = a.shape[0]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    load().orc_fft(_p(a), _p(_c(omega, 4)), c_uint32(log_n))
    return a


def dft_naive(a: np.ndarray, omega: np.ndarray) -> np.ndarray:
    a = _c(a, 4).reshape(-1, 4)
    n = a.shape[0]
    out = np.zeros_like(a)
    load().orc_dft_naive(_p(out), _p(a), _p(_c(omega, 4)), c_uint32(n.bit_length() - 1))
    return out


def lagrange_to_coeff(d: Domain, a: np.ndarray) -> np.ndarray:
    a = _c(a, 4).reshape(-1, 4).copy()
    load().orc_lagrange_to_coeff(ctypes.byref(d), _p(a))
    return a


def coeff_to_extended(d: Domain, coeffs: np.ndarray) -> np.ndarray:
    coeffs = _c(coeffs, 4).reshape(-1, 4)
    out = np.zeros((int(d.extended_n), 4), np.uint64)
    load().orc_coeff_to_extended(ctypes.byref(d), _p(out), _p(coeffs))
    return out


def extended_to_coeff(d: Domain, evals: np.ndarray) -> np.ndarray:
    evals = _c(evals, 4).reshape(-1, 4).copy()
    out = np.zeros((int(d.n) * int(d.quotient_poly_degree), 4), np.uint64)
    load().orc_extended_to_coeff(ctypes.byref(d), _p(out), _p(evals))
    return out


# ---------------------------------------------------------------- create_proof / verify_proof
class _Pk(ctypes.Structure):
    _fields_ = [("cs", c_void_p), ("fixed_values", c_void_p), ("sigma_values", c_void_p),
                ("params", c_void_p), ("vk_repr", c_uint64 * 4), ("derived", c_void_p)]


class Trace(ctypes.Structure):
    _fields_ = ([(name, c_void_p) for name in ("h_ext", "perm_z", "lookup_z", "permuted_input",
                                               "permuted_table", "h_pieces")]
                + [(name, c_uint64 * 4) for name in ("theta", "beta", "gamma", "y", "x", "v")]
                + [("n_sets", c_uint32)])

    def array(self, name: str, count: int) -> np.ndarray:
        ptr = getattr(self, name)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(c_uint64)), shape=(count, 4)).copy()

    def fe(self, name: str) -> np.ndarray:
        return np.array(list(getattr(self, name)), dtype=np.uint64)


class ProvingKey:
    """orc_pk: circuit image (harness/circuit.py CircuitImage), fixed/sigma values, SRS.  The rest of what keygen_pk
    stores (polys, cosets, l0 / l_last / l_active_row) is derived at the first create_proof and kept with the key;
    derive=False computes it inside every proof instead (same bytes: tests/test_oracle_prover.py)."""

    def __init__(self, image, fixed_values: np.ndarray, sigma_values: np.ndarray, params: Params,
                 vk_repr: np.ndarray, derive: bool = True):
        self.keep_derived = derive
        self.image = image
        self.fixed = np.ascontiguousarray(fixed_values, dtype=np.uint64)
        self.sigma = np.ascontiguousarray(sigma_values, dtype=np.uint64)
        self.params = params
        self.vk_repr = np.ascontiguousarray(vk_repr, dtype=np.uint64)
        self.c = _Pk()
        self.c.cs = ctypes.cast(image.ptr(), c_void_p)
        self.c.fixed_values = self.fixed.ctypes.data
        self.c.sigma_values = self.sigma.ctypes.data
        self.c.params = ctypes.cast(ctypes.byref(params), c_void_p)
        for i in range(4):
            self.c.vk_repr[i] = int(self.vk_repr[i])
        self.c.derived = None

    def derive(self):
        if self.keep_derived and not self.c.derived:
            load().orc_pk_derive(ctypes.byref(self.c))

    def __del__(self):
        try:
            if self.c.derived:
                load().orc_pk_release(ctypes.byref(self.c))
        except Exception:  # interpreter shutdown: the process's memory goes with it
            pass


def proof_size(image) -> int:
    f = load().orc_proof_size
    f.restype = c_size_t
    return int(f(image.ptr()))


def create_proof(pk: ProvingKey, advice: np.ndarray, instance: np.ndarray, seed, want_trace=False):
    """Returns (status, proof bytes, Trace | None)."""
    advice = np.ascontiguousarray(advice, dtype=np.uint64)
    instance = np.ascontiguousarray(instance, dtype=np.uint64)
    inst_len = instance.shape[1] if instance.ndim == 3 and instance.shape[0] else 0
    cap = proof_size(pk.image)
    buf = (ctypes.c_uint8 * cap)()
    plen = c_size_t(0)
    tr = Trace() if want_trace else None
    pk.derive()
    st = load().orc_create_proof(ctypes.byref(pk.c), _p(advice), _p(instance), c_size_t(inst_len),
                                 rng_key(seed), buf, c_size_t(cap), ctypes.byref(plen),
                                 ctypes.byref(tr) if tr is not None else None)
    return st, bytes(buf[: plen.value]), tr


def last_phase_ms() -> list:
    """Per-phase wall-clock of this thread's last create_proof (slots of zg_prover_phase_ms)."""
    out = (ctypes.c_double * 8)()
    load().orc_last_phase_ms(out, c_size_t(8))
    return list(out)


def verify_proof(pk: ProvingKey, instance: np.ndarray, proof: bytes) -> int:
    instance = np.ascontiguousarray(instance, dtype=np.uint64)
    inst_len = instance.shape[1] if instance.ndim == 3 and instance.shape[0] else 0
    buf = (ctypes.c_uint8 * len(proof)).from_buffer_copy(proof)
    return int(load().orc_verify_proof(ctypes.byref(pk.c), _p(instance), c_size_t(inst_len), buf,
                                       c_size_t(len(proof))))


def verify_proof_pairing(pk: ProvingKey, instance: np.ndarray, proof: bytes) -> int:
    """KZG/GWC verification with the BN254 pairing (uses g2 / s_g2 only, never the toxic scalar)."""
    instance = np.ascontiguousarray(instance, dtype=np.uint64)
    inst_len = instance.shape[1] if instance.ndim == 3 and instance.shape[0] else 0
    buf = (ctypes.c_uint8 * len(proof)).from_buffer_copy(proof)
    return int(load().orc_verify_proof_pairing(ctypes.byref(pk.c), _p(instance), c_size_t(inst_len), buf,
                                               c_size_t(len(proof))))


def trace_free(tr: Trace):
    load().orc_trace_free(ctypes.byref(tr))


def keccak256(data: bytes) -> bytes:
    out = (ctypes.c_uint8 * 32)()
    load().orc_keccak256(data, c_size_t(len(data)), out)
    return bytes(out)


def rng_key(seed) -> bytes:
    """The 32-byte blinding key: bytes pass through, an int (tests) becomes its little-endian encoding."""
    if isinstance(seed, (bytes, bytearray)):
        assert len(seed) == 32
        return bytes(seed)
    return int(seed).to_bytes(32, "little")


def rand_fr(seed, tag: int, index: int) -> np.ndarray:
    out = np.zeros(4, np.uint64)
    load().orc_rand_fr(_p(out), rng_key(seed), c_uint32(tag), c_uint64(index))
    return out


def chacha20_block(key: bytes, counter: int, nonce) -> bytes:
    out = (ctypes.c_uint32 * 16)()
    load().orc_chacha20_block(key, c_uint32(counter), (ctypes.c_uint32 * 3)(*nonce), out)
    return b"".join(int(w).to_bytes(4, "little") for w in out)


def grand_product(num: np.ndarray, den: np.ndarray, z0: np.ndarray) -> np.ndarray:
    num, den = _c(num, 4).reshape(-1, 4), _c(den, 4).reshape(-1, 4)
    z = np.zeros_like(num)
    load().orc_grand_product(_p(z), _p(num), _p(den), _p(_c(z0, 4)), c_size_t(num.shape[0]))
    return z


def eval_poly(coeffs: np.ndarray, x: np.ndarray) -> np.ndarray:
    coeffs = _c(coeffs, 4).reshape(-1, 4)
    out = np.zeros(4, np.uint64)
    load().orc_eval_poly(_p(out), _p(coeffs), c_size_t(coeffs.shape[0]), _p(_c(x, 4)))
    return out


def kate_division(a: np.ndarray, z: np.ndarray) -> np.ndarray:
    a = _c(a, 4).reshape(-1, 4)
    q = np.zeros_like(a)  # q[n-1] stays 0
    load().orc_kate_division(_p(q), _p(a), c_size_t(a.shape[0]), _p(_c(z, 4)))
    return q
