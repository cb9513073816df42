"""ctypes binding of libzg_halo2.so (include/zg_halo2.h) for the Python test/bench harness.

Field elements travel as numpy ``uint64`` arrays of shape ``(..., 4)`` (little-endian limbs,
Montgomery form: the memory format of halo2curves ``bn256::Fr`` / ``Fq``), affine points as
``(..., 8)`` and Jacobian points as ``(..., 12)``.

There is no CPU fallback: ``load()`` raises if the shared library is missing and ``Ctx()`` raises
if no HIP device is visible.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_int, c_size_t, c_uint32, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZG_HALO2_LIB") or os.path.join(_HERE, "libzg_halo2.so")  # (the override is for A/B builds)

FR_MODULUS = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
FQ_MODULUS = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47

_lib = None


class ZgError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"zg_halo2 status {status}: {msg}")
        self.status = status


def load() -> ctypes.CDLL:
    """Loads libzg_halo2.so; raises loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C 0g-halo2_amd` "
            "(or __graft_entry__.build()). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.zg_last_error.restype = c_char_p
    lib.zg_version.restype = c_char_p
    lib.zg_ctx_stream.restype = c_void_p
    lib.zg_ctx_stream.argtypes = [c_void_p]
    lib.zg_bases_len.restype = c_size_t
    lib.zg_bases_len.argtypes = [c_void_p]
    lib.zg_bases_window_bits.restype = c_uint32
    lib.zg_bases_window_bits.argtypes = [c_void_p]
    lib.zg_ctx_destroy.argtypes = [c_void_p]
    lib.zg_ctx_destroy.restype = None
    lib.zg_bases_free.argtypes = [c_void_p]
    lib.zg_bases_free.restype = None
    _lib = lib
    return lib


# Every symbol include/zg_halo2.h declares (tests check the shared object exports all of them).
ABI_SYMBOLS = [
    "zg_last_error", "zg_version", "zg_ctx_create", "zg_ctx_destroy", "zg_ctx_sync", "zg_ctx_stream",
    "zg_bases_register", "zg_bases_register_dev", "zg_bases_free", "zg_bases_len",
    "zg_bases_window_bits", "zg_msm", "zg_msm_batch", "zg_msm_batch_dev", "zg_msm_finish", "zg_g1_sum",
    "zg_ntt", "zg_intt", "zg_ntt_batch", "zg_intt_batch", "zg_ntt_batch_dev", "zg_coeff_to_extended",
    "zg_coeff_to_extended_batch_dev", "zg_extended_to_coeff", "zg_extended_to_coeff_dev",
    "zg_domain_omega", "zg_ctx_profile_enable", "zg_ctx_profile_collect", "zg_params_new",
    "zg_params_new_dev", "zg_prover_create", "zg_prover_destroy", "zg_prover_prove", "zg_prover_prove_dev",
    "zg_prover_proof_size", "zg_prover_fetch", "zg_grand_product_dev", "zg_eval_polys_dev",
    "zg_kate_division_dev", "zg_keccak256", "zg_ctx_profile_filter", "zg_prover_phase_ms", "zg_prover_gate_stats", "zg_ctx_trim", "zg_witness_plan_info", "zg_prover_set_overlap", "zg_ctx_set_msm_latency",
    "zg_prover_create_shared", "zg_prover_fork", "zg_prover_set_batch", "zg_prover_batch", "zg_prover_advice_slot",
    "zg_prover_prove_batch", "zg_prover_prove_batch_dev", "zg_prover_set_shard", "zg_prover_fetch_slot",
    "zg_grand_product", "zg_xyzz_sum_ranks", "zg_prover_evaluate_h",
    "zg_witness_plan_create", "zg_witness_plan_destroy", "zg_witness_plan_image_bytes", "zg_witness_plan_instance_len",
    "zg_witness_run_dev", "zg_prover_prove_images", "zg_prover_set_shard_rccl", "zg_xyzz_sum_ranks_dev", "zg_bases_enable_bit_table",
    "zg_tuning_set", "zg_tuning_get", "zg_tuning_names", "zg_bases_enable_digit_table", "zg_prover_enable_digit_tables",
]

def tuning_names() -> list:
    """Names of the library's tuning knobs (include/zg_halo2.h, "tuning")."""
    lib = load()
    lib.zg_tuning_names.restype = c_size_t
    n = int(lib.zg_tuning_names(None, c_size_t(0)))
    arr = (c_char_p * n)()
    lib.zg_tuning_names(arr, c_size_t(n))
    return [a.decode() for a in arr]


def tuning_set(name: str, value: int) -> None:
    """value < 0 restores the default.  Knobs that shape a proving key are read when a prover is created."""
    _check(load().zg_tuning_set(name.encode(), c_int(value)))


def tuning_get(name: str) -> int:
    v = c_int(0)
    _check(load().zg_tuning_get(name.encode(), ctypes.byref(v)))
    return v.value


EXCHANGE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_void_p)


def rng_key(seed) -> bytes:
    """The 32-byte blinding key of a proof: bytes pass through; an int (tests only) becomes its little-endian,
    zero-padded encoding.  Production callers pass 32 bytes from a CSPRNG (os.urandom(32))."""
    if isinstance(seed, (bytes, bytearray)):
        assert len(seed) == 32
        return bytes(seed)
    return int(seed).to_bytes(32, "little")


class KernelStat(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("launches", ctypes.c_uint64), ("total_ms", ctypes.c_double),
                ("algo_bytes", ctypes.c_double), ("unit_bytes", ctypes.c_double)]


def _check(status: int) -> None:
    if status != 0:
        raise ZgError(status, load().zg_last_error().decode())


def _fr(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.shape[-1] == 4, a.shape
    return a


def _ptr(a: np.ndarray) -> c_void_p:
    return c_void_p(a.ctypes.data)


def int_to_limbs(x: int) -> np.ndarray:
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def limbs_to_int(a) -> int:
    return sum(int(a[i]) << (64 * i) for i in range(4))


def fr_from_int(x: int) -> np.ndarray:
    """Canonical integer -> Montgomery limbs (host helper for tests)."""
    return int_to_limbs((x % FR_MODULUS) * (1 << 256) % FR_MODULUS)


def fr_to_int(a) -> int:
    return limbs_to_int(a) * pow(1 << 256, -1, FR_MODULUS) % FR_MODULUS


def fq_from_int(x: int) -> np.ndarray:
    return int_to_limbs((x % FQ_MODULUS) * (1 << 256) % FQ_MODULUS)


def fq_to_int(a) -> int:
    return limbs_to_int(a) * pow(1 << 256, -1, FQ_MODULUS) % FQ_MODULUS


def domain_omega(log_n: int):
    om = np.zeros(4, np.uint64)
    omi = np.zeros(4, np.uint64)
    _check(load().zg_domain_omega(c_uint32(log_n), _ptr(om), _ptr(omi)))
    return om, omi


def g1_sum(parts: np.ndarray) -> np.ndarray:
    parts = np.ascontiguousarray(parts, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros(12, np.uint64)
    _check(load().zg_g1_sum(_ptr(parts), c_size_t(parts.shape[0]), _ptr(out)))
    return out


def xyzz_sum_ranks(parts: np.ndarray) -> np.ndarray:
    """parts: uint64[world, count, 16] extended-Jacobian partial sums -> uint64[count, 12] normalised sums."""
    parts = np.ascontiguousarray(parts, dtype=np.uint64)
    world, count = parts.shape[0], parts.shape[1]
    out = np.zeros((count, 12), np.uint64)
    _check(load().zg_xyzz_sum_ranks(_ptr(parts), c_size_t(world), c_size_t(count), _ptr(out)))
    return out


import atexit
import weakref

_live_contexts = weakref.WeakSet()


@atexit.register
def _close_contexts_at_exit():
    # contexts (and their provers) must be torn down while the HIP runtime is still alive, not by the garbage
    # collector after the interpreter has started unloading libraries
    for c in list(_live_contexts):
        try:
            c.close()
        except Exception:  # noqa: BLE001
            pass


class Ctx:
    """One GPU (zg_ctx)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = c_void_p()
        _check(self.lib.zg_ctx_create(c_int(device), ctypes.byref(h)))
        self.h = h
        self._children = []  # weak references to the provers created on this context: they must go first
        _live_contexts.add(self)

    def _adopt(self, child):
        import weakref

        self._children.append(weakref.ref(child))

    def close(self):
        if getattr(self, "h", None):
            for ref in self._children:
                child = ref()
                if child is not None:
                    child.close()
            self._children = []
            self.lib.zg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(self.lib.zg_ctx_sync(self.h))

    def drop_workspace(self) -> int:
        """zg_ctx_trim: the free blocks of the workspace pool go back to the device allocator; returns the bytes freed"""
        freed = ctypes.c_uint64(0)
        _check(self.lib.zg_ctx_trim(self.h, ctypes.byref(freed)))
        return freed.value

    @property
    def stream(self) -> int:
        return self.lib.zg_ctx_stream(self.h)

    def profile(self, on: bool):
        _check(self.lib.zg_ctx_profile_enable(self.h, c_int(1 if on else 0)))

    def profile_filter(self, kernel_name: str | None):
        _check(self.lib.zg_ctx_profile_filter(self.h, kernel_name.encode() if kernel_name else None))

    def profile_collect(self) -> dict:
        """{kernel: (launches, total_ms, algo_bytes, unit_bytes)} since the last collect; synchronises.  algo_bytes = what
        the kernel itself streams, unit_bytes = SURVEY 8d's figure of the units it carries (include/zg_halo2.h)."""
        cap = 256  # (distinct ZG_LAUNCH labels: ~60; the call drains the records, so the array must hold them all at once)
        arr = (KernelStat * cap)()
        cnt = c_size_t(0)
        _check(self.lib.zg_ctx_profile_collect(self.h, arr, c_size_t(cap), ctypes.byref(cnt)))
        if cnt.value > cap:
            raise ZgError(-1, f"zg_ctx_profile_collect: {cnt.value} kernels for an array of {cap}: records were dropped")
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms), float(arr[i].algo_bytes), float(arr[i].unit_bytes))
                for i in range(cnt.value)}

    # ---- SRS ----
    def params_new(self, k: int, s: np.ndarray):
        """ParamsKZG::new(k) with toxic scalar s -> (g, g_lagrange) as uint64[n, 8] host arrays."""
        n = 1 << k
        g = np.zeros((n, 8), np.uint64)
        gl = np.zeros((n, 8), np.uint64)
        _check(self.lib.zg_params_new(self.h, c_uint32(k), _ptr(_fr(s)), _ptr(g), _ptr(gl)))
        return g, gl

    def params_new_dev(self, k: int, s: np.ndarray, d_g: int, d_gl: int):
        _check(self.lib.zg_params_new_dev(self.h, c_uint32(k), _ptr(_fr(s)), c_void_p(d_g), c_void_p(d_gl)))

    # ---- MSM ----
    def set_msm_latency(self, latency: bool):
        """True (default): two lanes per addition in the MSM reduction; False: one lane per addition."""
        _check(self.lib.zg_ctx_set_msm_latency(self.h, ctypes.c_int(1 if latency else 0)))

    def register_bases(self, bases: np.ndarray, window_bits: int = 0) -> "Bases":
        bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
        h = c_void_p()
        _check(self.lib.zg_bases_register(self.h, _ptr(bases), c_size_t(bases.shape[0]),
                                          c_uint32(window_bits), ctypes.byref(h)))
        return Bases(self, h)

    def enable_bit_table(self, bases: "Bases", digit_width: int):
        """Bit-position table + free-position odd digits for this context's throughput-form MSMs (zg_bases_enable_bit_table)."""
        _check(self.lib.zg_bases_enable_bit_table(self.h, bases.h, c_uint32(digit_width)))

    def enable_digit_table(self, bases: "Bases", window_bits: int = 0):
        """Every multiple of every window, for lone MSMs on a latency-form context (zg_bases_enable_digit_table)."""
        _check(self.lib.zg_bases_enable_digit_table(self.h, bases.h, c_uint32(window_bits)))

    def register_bases_dev(self, d_ptr: int, n: int, window_bits: int = 0) -> "Bases":
        h = c_void_p()
        _check(self.lib.zg_bases_register_dev(self.h, c_void_p(d_ptr), c_size_t(n), c_uint32(window_bits),
                                              ctypes.byref(h)))
        return Bases(self, h)

    def msm(self, bases: "Bases", scalars: np.ndarray) -> np.ndarray:
        scalars = _fr(scalars).reshape(-1, 4)
        out = np.zeros(12, np.uint64)
        _check(self.lib.zg_msm(self.h, bases.h, _ptr(scalars), c_size_t(scalars.shape[0]), _ptr(out)))
        return out

    def msm_batch(self, bases: "Bases", scalars: np.ndarray) -> np.ndarray:
        scalars = _fr(scalars)
        assert scalars.ndim == 3
        batch, n = scalars.shape[0], scalars.shape[1]
        ptrs = (c_void_p * batch)(*[scalars[b].ctypes.data for b in range(batch)])
        out = np.zeros((batch, 12), np.uint64)
        _check(self.lib.zg_msm_batch(self.h, bases.h, ptrs, c_size_t(batch), c_size_t(n), _ptr(out)))
        return out

    def msm_batch_dev(self, bases: "Bases", d_scalars: int, stride: int, batch: int, n: int, d_out: int):
        _check(self.lib.zg_msm_batch_dev(self.h, bases.h, c_void_p(d_scalars), c_size_t(stride),
                                         c_size_t(batch), c_size_t(n), c_void_p(d_out)))

    def msm_finish(self, d_xyzz: int, batch: int) -> np.ndarray:
        out = np.zeros((batch, 12), np.uint64)
        _check(self.lib.zg_msm_finish(self.h, c_void_p(d_xyzz), c_size_t(batch), _ptr(out)))
        return out

    # ---- NTT ----
    def ntt(self, a: np.ndarray, omega: np.ndarray, divisor: np.ndarray | None = None) -> np.ndarray:
        """Returns best_fft(a, omega) (times divisor when given); a is not modified."""
        a = _fr(a).reshape(-1, 4).copy()
        n = a.shape[0]
        log_n = n.bit_length() - 1
        assert 1 << log_n == n
        omega = _fr(omega)
        if divisor is None:
            _check(self.lib.zg_ntt(self.h, _ptr(a), c_uint32(log_n), _ptr(omega)))
        else:
            divisor = _fr(divisor)
            _check(self.lib.zg_intt(self.h, _ptr(a), c_uint32(log_n), _ptr(omega), _ptr(divisor)))
        return a

    def ntt_batch(self, a: np.ndarray, omega: np.ndarray, divisor: np.ndarray | None = None) -> np.ndarray:
        a = _fr(a).copy()
        assert a.ndim == 3
        batch, n = a.shape[0], a.shape[1]
        log_n = n.bit_length() - 1
        ptrs = (c_void_p * batch)(*[a[b].ctypes.data for b in range(batch)])
        omega = _fr(omega)
        if divisor is None:
            _check(self.lib.zg_ntt_batch(self.h, ptrs, c_size_t(batch), c_uint32(log_n), _ptr(omega)))
        else:
            divisor = _fr(divisor)
            _check(self.lib.zg_intt_batch(self.h, ptrs, c_size_t(batch), c_uint32(log_n), _ptr(omega),
                                          _ptr(divisor)))
        return a

    def ntt_batch_dev(self, d_a: int, stride: int, batch: int, log_n: int, omega: np.ndarray,
                      divisor: np.ndarray | None = None):
        omega = _fr(omega)
        dv = _ptr(_fr(divisor)) if divisor is not None else c_void_p(0)
        _check(self.lib.zg_ntt_batch_dev(self.h, c_void_p(d_a), c_size_t(stride), c_size_t(batch),
                                         c_uint32(log_n), _ptr(omega), dv))

    # ---- stand-alone prover building blocks (device pointers) ----
    def grand_product(self, num: np.ndarray, den: np.ndarray, z0: np.ndarray) -> np.ndarray:
        num, den = _fr(num).reshape(-1, 4), _fr(den).reshape(-1, 4)
        z = np.zeros_like(num)
        _check(self.lib.zg_grand_product(self.h, _ptr(num), _ptr(den), _ptr(_fr(z0)), c_size_t(num.shape[0]), _ptr(z)))
        return z

    def grand_product_dev(self, d_num: int, d_den: int, z0: np.ndarray, n: int, d_z: int):
        _check(self.lib.zg_grand_product_dev(self.h, c_void_p(d_num), c_void_p(d_den), _ptr(_fr(z0)), c_size_t(n),
                                             c_void_p(d_z)))

    def eval_polys_dev(self, d_polys: int, stride: int, n: int, poly_index, points: np.ndarray) -> np.ndarray:
        idx = np.ascontiguousarray(poly_index, dtype=np.uint32)
        points = _fr(points).reshape(-1, 4)
        out = np.zeros((idx.shape[0], 4), np.uint64)
        _check(self.lib.zg_eval_polys_dev(self.h, c_void_p(d_polys), c_size_t(stride), c_size_t(n),
                                          c_void_p(idx.ctypes.data), _ptr(points), c_size_t(idx.shape[0]), _ptr(out)))
        return out

    def kate_division_dev(self, d_a: int, n: int, z: np.ndarray, d_q: int):
        _check(self.lib.zg_kate_division_dev(self.h, c_void_p(d_a), c_size_t(n), _ptr(_fr(z)), c_void_p(d_q)))
        self.sync()

    def coeff_to_extended(self, coeffs: np.ndarray, k: int, ext_k: int) -> np.ndarray:
        coeffs = _fr(coeffs).reshape(-1, 4)
        assert coeffs.shape[0] == 1 << k
        out = np.zeros((1 << ext_k, 4), np.uint64)
        _check(self.lib.zg_coeff_to_extended(self.h, _ptr(coeffs), c_uint32(k), c_uint32(ext_k), _ptr(out)))
        return out

    def coeff_to_extended_batch_dev(self, d_in: int, in_stride: int, d_out: int, out_stride: int, batch: int,
                                    k: int, ext_k: int):
        _check(self.lib.zg_coeff_to_extended_batch_dev(self.h, c_void_p(d_in), c_size_t(in_stride),
                                                       c_void_p(d_out), c_size_t(out_stride), c_size_t(batch),
                                                       c_uint32(k), c_uint32(ext_k)))

    def extended_to_coeff(self, evals: np.ndarray, k: int, ext_k: int, out_len: int) -> np.ndarray:
        evals = _fr(evals).reshape(-1, 4).copy()
        assert evals.shape[0] == 1 << ext_k
        out = np.zeros((out_len, 4), np.uint64)
        _check(self.lib.zg_extended_to_coeff(self.h, _ptr(evals), c_uint32(k), c_uint32(ext_k),
                                             c_size_t(out_len), _ptr(out)))
        return out

    def extended_to_coeff_dev(self, d_evals: int, k: int, ext_k: int, out_len: int, d_out: int):
        _check(self.lib.zg_extended_to_coeff_dev(self.h, c_void_p(d_evals), c_uint32(k), c_uint32(ext_k),
                                                 c_size_t(out_len), c_void_p(d_out)))


def keccak256(data: bytes) -> bytes:
    out = (ctypes.c_uint8 * 32)()
    load().zg_keccak256(data, c_size_t(len(data)), out)
    return bytes(out)


class WitnessPlan:
    """zg_witness_plan: the recorded witness program of a circuit (harness/witness_tape.py WitnessProgram.arrays())
    on one context; run() writes the advice columns of a batch of inputs into device buffers."""

    def __init__(self, ctx: Ctx, arrays: dict):
        self.ctx = ctx
        lib = ctx.lib
        lib.zg_witness_plan_destroy.argtypes = [c_void_p]
        lib.zg_witness_plan_destroy.restype = None
        ops = np.ascontiguousarray(arrays["ops"], dtype=np.uint64)
        level_start = np.ascontiguousarray(arrays["level_start"], dtype=np.uint32)
        consts = np.ascontiguousarray(arrays["consts"], dtype=np.uint64)
        table = np.ascontiguousarray(arrays["table"], dtype=np.uint64)
        cell_slot = np.ascontiguousarray(arrays["cell_slot"], dtype=np.uint32)
        inst = np.ascontiguousarray(arrays["instance_slots"], dtype=np.uint32)
        self.n_advice, n = cell_slot.shape
        self.k = n.bit_length() - 1
        self.n_instance = int(inst.shape[0])
        self.image_bytes = int(arrays["image_bytes"])
        h = c_void_p()
        _check(lib.zg_witness_plan_create(ctx.h, _ptr(ops), c_size_t(ops.shape[0]), _ptr(level_start),
                                          c_size_t(level_start.shape[0] - 1), _ptr(consts), c_size_t(consts.shape[0]),
                                          _ptr(table), c_size_t(table.shape[0]), _ptr(cell_slot), c_uint32(self.n_advice),
                                          c_uint32(self.k), _ptr(inst), c_size_t(self.n_instance),
                                          c_size_t(self.image_bytes), ctypes.byref(h)))
        self.h = h
        ctx._adopt(self)

    def info(self) -> dict:
        """zg_witness_plan_info: how the program was laid out (LDS cells, values left in HBM, levels with a global barrier)"""
        out = (ctypes.c_uint64 * 10)()
        _check(self.ctx.lib.zg_witness_plan_info(self.h, out, c_size_t(10)))
        return dict(zip(("lds_bytes", "narrow_cells", "wide_cells", "values_in_lds", "values_in_hbm", "hbm_levels", "wide_values", "levels",
                         "narrow_ops", "lanes"),
                        (int(x) for x in out)))

    def run(self, images: np.ndarray, d_advice) -> np.ndarray:
        """images: uint8[count, image_bytes...]; d_advice: `count` device addresses ([n_advice][2^k] field elements
        each).  Returns the instance values, uint64[count, n_instance, 4] (Montgomery form)."""
        images = np.ascontiguousarray(images, dtype=np.uint8).reshape(len(d_advice), -1)
        assert images.shape[1] == self.image_bytes, "image size does not match the recorded program"
        count = images.shape[0]
        ptrs = (c_void_p * count)(*[c_void_p(a) for a in d_advice])
        out = np.zeros((count, self.n_instance, 4), np.uint64)
        _check(self.ctx.lib.zg_witness_run_dev(self.h, _ptr(images), c_size_t(count), ptrs, _ptr(out)))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.zg_witness_plan_destroy(self.h)
            self.h = None


class Prover:
    """zg_prover: create_proof for one circuit (circuit.py CircuitImage) on one GPU, one proof or a lock-step batch."""

    def __init__(self, ctx: Ctx, image, fixed_values: np.ndarray = None, sigma_values: np.ndarray = None, g=None,
                 g_lagrange=None, vk_repr: np.ndarray = None, _handle=None, _keep=None):
        """g / g_lagrange: uint64[n, 8] host arrays, or two `Bases` handles shared between provers."""
        self.ctx = ctx
        self.image = image
        lib = ctx.lib
        lib.zg_prover_proof_size.restype = c_size_t
        lib.zg_prover_proof_size.argtypes = [c_void_p]
        lib.zg_prover_destroy.argtypes = [c_void_p]
        lib.zg_prover_destroy.restype = None
        lib.zg_prover_batch.restype = c_size_t
        lib.zg_prover_batch.argtypes = [c_void_p]
        lib.zg_prover_advice_slot.restype = c_void_p
        lib.zg_prover_advice_slot.argtypes = [c_void_p, c_size_t]
        self._keep = _keep  # (a fork holds its parent: the parent's exchange trampolines outlive every fork)
        self._exchange = None
        # every exchange trampoline this prover was ever given: zg_prover_fork copies the raw function pointer into the
        # fork on the C side, so a superseded callback must stay alive for as long as a fork may still call it
        self._exchanges = [] if _keep is None else _keep._exchanges
        if _handle is not None:
            h = _handle
        else:
            fixed_values = np.ascontiguousarray(fixed_values, dtype=np.uint64)
            sigma_values = np.ascontiguousarray(sigma_values, dtype=np.uint64)
            h = c_void_p()
            if isinstance(g, Bases):
                self._bases = (g, g_lagrange)  # keep the shared tables alive
                _check(lib.zg_prover_create_shared(ctx.h, image.ptr(), _ptr(fixed_values), _ptr(sigma_values), g.h,
                                                   g_lagrange.h, _ptr(_fr(vk_repr)), ctypes.byref(h)))
            else:
                g = np.ascontiguousarray(g, dtype=np.uint64)
                g_lagrange = np.ascontiguousarray(g_lagrange, dtype=np.uint64)
                _check(lib.zg_prover_create(ctx.h, image.ptr(), _ptr(fixed_values), _ptr(sigma_values), _ptr(g),
                                            _ptr(g_lagrange), _ptr(_fr(vk_repr)), ctypes.byref(h)))
        self.h = h
        ctx._adopt(self)
        self.n = 1 << image.c.k
        self.n_advice = image.c.n_advice
        self.proof_cap = int(lib.zg_prover_proof_size(h))

    def fork(self, ctx: Ctx) -> "Prover":
        """Another prover on the same proving key and base tables (no copy), on another context of the device."""
        h = c_void_p()
        _check(self.ctx.lib.zg_prover_fork(self.h, ctx.h, ctypes.byref(h)))
        return Prover(ctx, self.image, _handle=h, _keep=self)

    def set_batch(self, max_batch: int):
        _check(self.ctx.lib.zg_prover_set_batch(self.h, c_size_t(max_batch)))

    @property
    def batch(self) -> int:
        return int(self.ctx.lib.zg_prover_batch(self.h))

    def advice_slot(self, slot: int) -> int:
        """Device address of the advice columns of proof slot `slot` ([n_advice][2^k] field elements)."""
        return int(self.ctx.lib.zg_prover_advice_slot(self.h, c_size_t(slot)))

    def set_shard(self, rank: int, world: int, first_point: int, exchange):
        """exchange(send: bytes-like view, recv: writable view of world * len(send) bytes) = all-gather."""
        def _cb(_user, send, nbytes, recv):
            try:
                src = (ctypes.c_uint8 * nbytes).from_address(send)
                dst = (ctypes.c_uint8 * (nbytes * world)).from_address(recv)
                exchange(src, dst)
                return 0
            except Exception:  # noqa: BLE001 -- nothing may propagate through the C frames
                import traceback

                traceback.print_exc()
                return 1

        self._exchange = EXCHANGE_FN(_cb) if world > 1 else None
        if self._exchange is not None:
            self._exchanges.append(self._exchange)
        fn = self._exchange if self._exchange is not None else ctypes.cast(None, EXCHANGE_FN)
        _check(self.ctx.lib.zg_prover_set_shard(self.h, c_uint32(rank), c_uint32(world), c_size_t(first_point), fn, None))

    def set_shard_c(self, rank: int, world: int, first_point: int, fn_ptr: int, user: int = 0):
        """zg_prover_set_shard with a C function as the exchange (address of an
        `int fn(void *user, const void *send, size_t nbytes, void *recv)`): no Python in the provers' threads."""
        _check(self.ctx.lib.zg_prover_set_shard(self.h, c_uint32(rank), c_uint32(world), c_size_t(first_point),
                                                ctypes.cast(fn_ptr, EXCHANGE_FN), c_void_p(user)))

    def set_shard_rccl(self, rank: int, world: int, first_point: int, comm: int):
        """comm: an initialised ncclComm_t (multi_gpu.RcclComm(...).handle); the exchange then runs inside the library."""
        _check(self.ctx.lib.zg_prover_set_shard_rccl(self.h, c_uint32(rank), c_uint32(world), c_size_t(first_point),
                                                     c_void_p(comm)))

    @staticmethod
    def _inst(instance):
        instance = np.ascontiguousarray(instance, dtype=np.uint64)
        inst_len = instance.shape[1] if instance.ndim == 3 and instance.shape[0] else 0
        return instance, inst_len

    def prove(self, advice: np.ndarray, instance: np.ndarray, seed) -> bytes:
        advice = np.ascontiguousarray(advice, dtype=np.uint64)
        instance, inst_len = self._inst(instance)
        buf = (ctypes.c_uint8 * self.proof_cap)()
        plen = c_size_t(0)
        _check(self.ctx.lib.zg_prover_prove(self.h, _ptr(advice), _ptr(instance), c_size_t(inst_len), rng_key(seed), buf,
                                            c_size_t(self.proof_cap), ctypes.byref(plen)))
        return bytes(buf[: plen.value])

    def prove_dev(self, d_advice: int, instance: np.ndarray, seed) -> bytes:
        instance, inst_len = self._inst(instance)
        buf = (ctypes.c_uint8 * self.proof_cap)()
        plen = c_size_t(0)
        _check(self.ctx.lib.zg_prover_prove_dev(self.h, c_void_p(d_advice), _ptr(instance), c_size_t(inst_len),
                                                rng_key(seed), buf, c_size_t(self.proof_cap), ctypes.byref(plen)))
        return bytes(buf[: plen.value])

    def prove_batch(self, advice, instances, seeds, device=False, raise_on_error=True):
        """count = len(seeds) proofs in lock step.  advice: list of host arrays (device=False) or device addresses
        (device=True); None entries (or advice=None) = the slot already holds the columns.  Returns (proofs, statuses)."""
        count = len(seeds)
        lib = self.ctx.lib
        keep = []
        if advice is None:
            adv_ptrs = None
        elif device:
            adv_ptrs = (c_void_p * count)(*[c_void_p(a) if a else None for a in advice])
        else:
            keep = [np.ascontiguousarray(a, dtype=np.uint64) if a is not None else None for a in advice]
            adv_ptrs = (c_void_p * count)(*[c_void_p(a.ctypes.data) if a is not None else None for a in keep])
        insts, inst_len = [], 0
        for b in range(count):
            i, inst_len = self._inst(instances[b])
            insts.append(i)
        inst_ptrs = (c_void_p * count)(*[c_void_p(i.ctypes.data) for i in insts])
        keys = b"".join(rng_key(s) for s in seeds)
        bufs = [(ctypes.c_uint8 * self.proof_cap)() for _ in range(count)]
        out_ptrs = (c_void_p * count)(*[ctypes.addressof(b) for b in bufs])
        lens = (c_size_t * count)()
        sts = (c_int * count)()
        fn = lib.zg_prover_prove_batch_dev if device else lib.zg_prover_prove_batch
        st = fn(self.h, c_size_t(count), adv_ptrs, inst_ptrs, c_size_t(inst_len), keys, out_ptrs,
                c_size_t(self.proof_cap), lens, sts)
        if st != 0 and raise_on_error:
            _check(st)
        return [bytes(bufs[b][: lens[b]]) for b in range(count)], list(sts)

    def prove_images(self, plan: "WitnessPlan", images: np.ndarray, seeds, raise_on_error=True):
        """Wnn::proof for a batch: image bytes -> (proofs, outputs uint64[count, n_instance, 4], statuses)."""
        count = len(seeds)
        images = np.ascontiguousarray(images, dtype=np.uint8).reshape(count, -1)
        keys = b"".join(rng_key(s) for s in seeds)
        bufs = [(ctypes.c_uint8 * self.proof_cap)() for _ in range(count)]
        out_ptrs = (c_void_p * count)(*[ctypes.addressof(b) for b in bufs])
        lens = (c_size_t * count)()
        sts = (c_int * count)()
        outputs = np.zeros((count, plan.n_instance, 4), np.uint64)
        st = self.ctx.lib.zg_prover_prove_images(self.h, plan.h, _ptr(images), c_size_t(count), keys, out_ptrs,
                                                 c_size_t(self.proof_cap), lens, _ptr(outputs), sts)
        if st != 0 and raise_on_error:
            _check(st)
        return [bytes(bufs[b][: lens[b]]) for b in range(count)], outputs, list(sts)

    def evaluate_h(self, advice_polys, instance_polys, perm_z_polys, lookup_z_polys, permuted_polys, theta, beta, gamma, y,
                   extended_n: int) -> np.ndarray:
        """Evaluator::evaluate_h / (X^n - 1) from coefficient forms (uint64[count, n, 4] each) -> uint64[extended_n, 4]."""
        arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in (advice_polys, instance_polys, perm_z_polys, lookup_z_polys,
                                                                    permuted_polys)]
        ptrs = [_ptr(a) if a.size else None for a in arrs]
        out = np.zeros((extended_n, 4), np.uint64)
        _check(self.ctx.lib.zg_prover_evaluate_h(self.h, *ptrs, _ptr(_fr(theta)), _ptr(_fr(beta)), _ptr(_fr(gamma)),
                                                 _ptr(_fr(y)), _ptr(out)))
        return out

    def set_overlap(self, enable, digit_tables: bool = False):
        """True (default): transforms on a side stream (latency); False: one stream per proof (throughput).
        enable == "tables" or digit_tables=True: the latency form AND its digit tables (enable_digit_tables) -- the
        tables are never built implicitly."""
        if enable == "tables":
            enable, digit_tables = True, True
        _check(self.ctx.lib.zg_prover_set_overlap(self.h, ctypes.c_int(1 if enable else 0)))
        if enable and digit_tables:
            self.enable_digit_tables()

    def enable_digit_tables(self, max_bytes: int = 0) -> int:
        """zg_prover_enable_digit_tables: the lone-proof tables of the prover's three base sets (78 GB at k = 14), built NOW;
        max_bytes 0 = the library's cap.  Returns the bytes resident afterwards (0: none fit / none at this size)."""
        built = ctypes.c_uint64(0)
        _check(self.ctx.lib.zg_prover_enable_digit_tables(self.h, ctypes.c_uint64(max_bytes), ctypes.byref(built)))
        return built.value

    def phase_ms(self) -> list:
        out = (ctypes.c_double * 8)()
        _check(self.ctx.lib.zg_prover_phase_ms(self.h, out, c_size_t(8)))
        return list(out)

    def gate_stats(self) -> dict:
        """zg_prover_gate_stats: what the gate (ZG_LAT_GATE) did on this prover so far"""
        out = (ctypes.c_uint64 * 4)()
        _check(self.ctx.lib.zg_prover_gate_stats(self.h, out, c_size_t(4)))
        return dict(zip(("gated_proofs", "gates_armed", "remade_plain", "yields"), (int(x) for x in out)))

    def fetch(self, what: int, index: int, count: int, slot: int = 0) -> np.ndarray:
        out = np.zeros((count, 4), np.uint64)
        _check(self.ctx.lib.zg_prover_fetch_slot(self.h, c_size_t(slot), c_uint32(what), c_uint32(index), _ptr(out),
                                                 c_size_t(count)))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.zg_prover_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Bases:
    def __init__(self, ctx: Ctx, h: c_void_p):
        self.ctx = ctx
        self.h = h

    def __len__(self):
        return self.ctx.lib.zg_bases_len(self.h)

    @property
    def window_bits(self) -> int:
        return self.ctx.lib.zg_bases_window_bits(self.h)

    def free(self):
        if self.h:
            self.ctx.lib.zg_bases_free(self.h)
            self.h = None
