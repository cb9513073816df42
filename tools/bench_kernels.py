#!/usr/bin/env python3
"""Stand-alone MSM and NTT entry points of the C ABI on SURVEY.md 8(d)'s synthetic inputs (run on the GPU box):

    python tools/bench_kernels.py [out.json]

  MSM   zg_msm_batch_dev over ParamsKZG::g_lagrange of n = 2^14, 2^15, 2^17; scalar vectors resident in HBM:
        (A) uniform Fr, (B) "advice-like" (70 % zero, 20 % in {0, 1}, 8 % below 2^8, 2 % uniform); batch 1, 6, 20, 30;
        the throughput form (strip reduction; bit-position tables for width-15 odd digits on the uniform vectors) and the
        latency form (window tables, several lanes per EC addition; digit tables where the prover would build them).
  NTT   zg_ntt_batch_dev (the inverse transform with its divisor) at log n = 14, 15, 17, 18, 20 and
        zg_coeff_to_extended_batch_dev n -> 8n at k = 14, 15, 17; batch 1, 7, 21.
Per case: microseconds per call (device time of its kernels from per-launch HIP events AND host wall time around
synchronised calls), units per second, ALGORITHMIC bytes (an MSM n * 96 + 96 B, a transform (in + out) * 32 B)
per second against the 8 TB/s HBM peak.  Results are not checked here (tests/test_gpu_msm.py, tests/test_gpu_ntt.py do)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the library: one HIP runtime per process)
import zg_halo2 as zg  # noqa: E402

HBM_PEAK = 8.0e12
gen = np.random.default_rng(20241004)


def uniform(shape):
    a = gen.integers(0, 1 << 62, size=shape + (4,), dtype=np.int64).astype(np.uint64)
    a[..., 3] &= np.uint64((1 << 60) - 1)  # (any value below r is a field element in Montgomery form)
    return a


def advice_like(shape):
    """the mix of a WnnCircuit advice column"""
    a = np.zeros(shape + (4,), np.uint64)
    u = gen.random(shape)
    one = zg.fr_from_int(1)
    a[(u >= 0.70) & (u < 0.80)] = one
    small = (u >= 0.90) & (u < 0.98)
    vals = np.stack([zg.fr_from_int(int(v)) for v in range(256)])
    a[small] = vals[gen.integers(0, 256, size=int(small.sum()))]
    big = u >= 0.98
    a[big] = uniform((int(big.sum()),))
    return a


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def timed(ctx, fn, reps):
    """(device seconds per call from the launches' own events, wall seconds per call, kernels' share)"""
    fn()
    fn()
    ctx.sync()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    wall = (time.perf_counter() - t0) / reps
    st = ctx.profile_collect()
    ctx.profile(False)
    device = sum(v[1] for v in st.values()) / reps * 1e-3
    return device, wall, {k_: round(v[1] / reps * 1e3, 1) for k_, v in sorted(st.items(), key=lambda kv: -kv[1][1])}


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    ctx = zg.Ctx(0)
    rows = []
    s = zg.fr_from_int(0x5EED)
    for k in (14, 15, 17):
        n = 1 << k
        g, gl = ctx.params_new(k, s)
        for form in ("throughput", "latency"):
            bases = ctx.register_bases(gl)
            ctx.set_msm_latency(form == "latency")
            if form == "throughput":
                ctx.enable_bit_table(bases, 15 if k <= 15 else 16)
            elif k <= 15:
                ctx.enable_digit_table(bases)
            for dist, make in (("uniform", uniform), ("advice-like", advice_like)):
                for batch in (1, 6, 20, 30):
                    d_s = dev(make((batch, n)))
                    d_out = torch.empty((batch, 16), dtype=torch.int64, device="cuda")
                    dt, wall, kern = timed(ctx, lambda: ctx.msm_batch_dev(bases, d_s.data_ptr(), n, batch, n, d_out.data_ptr()),
                                           10 if batch * n <= (1 << 20) else 4)
                    by = batch * (n * 96 + 96)
                    rows.append({"op": "msm", "form": form, "k": k, "scalars": dist, "batch": batch,
                                 "device_us": round(dt * 1e6, 1), "wall_us": round(wall * 1e6, 1),
                                 "msm_per_s": round(batch / wall, 1), "pairs_per_s": batch * n / wall,
                                 "algo_GBps": round(by / dt / 1e9, 1), "frac_of_hbm_peak": round(by / dt / HBM_PEAK, 5),
                                 "kernels_us": kern})
                    print(rows[-1], flush=True)
                    del d_s, d_out
            bases.free()
    ctx.set_msm_latency(False)
    for log_n in (14, 15, 17, 18, 20):
        n = 1 << log_n
        om, omi = zg.domain_omega(log_n)
        div = zg.fr_from_int(pow(n, -1, zg.FR_MODULUS))
        for batch in (1, 7, 21):
            d_a = dev(uniform((batch, n)))
            dt, wall, kern = timed(ctx, lambda: ctx.ntt_batch_dev(d_a.data_ptr(), n, batch, log_n, omi, div), 10)
            by, bf = batch * 2 * n * 32, batch * (n // 2) * log_n
            rows.append({"op": "intt", "log_n": log_n, "batch": batch, "device_us": round(dt * 1e6, 1), "wall_us": round(wall * 1e6, 1),
                         "butterflies_per_s": bf / dt, "algo_GBps": round(by / dt / 1e9, 1),
                         "frac_of_hbm_peak": round(by / dt / HBM_PEAK, 5), "kernels_us": kern})
            print(rows[-1], flush=True)
            del d_a
    for k in (14, 15, 17):
        n, ek = 1 << k, k + 3
        en = 1 << ek
        for batch in (1, 7, 21):
            d_in = dev(uniform((batch, n)))
            d_out = torch.empty((batch, en, 4), dtype=torch.int64, device="cuda")
            dt, wall, kern = timed(ctx, lambda: ctx.coeff_to_extended_batch_dev(d_in.data_ptr(), n, d_out.data_ptr(), en, batch, k, ek), 6)
            by, bf = batch * (n + en) * 32, batch * (en // 2) * ek
            rows.append({"op": "coeff_to_extended", "k": k, "ext_k": ek, "batch": batch, "device_us": round(dt * 1e6, 1),
                         "wall_us": round(wall * 1e6, 1), "butterflies_per_s": bf / dt, "algo_GBps": round(by / dt / 1e9, 1),
                         "frac_of_hbm_peak": round(by / dt / HBM_PEAK, 5), "kernels_us": kern})
            print(rows[-1], flush=True)
            del d_in, d_out
    ctx.close()
    doc = {"_note": "tools/bench_kernels.py: stand-alone zg_msm_batch_dev / zg_ntt_batch_dev / zg_coeff_to_extended_batch_dev on "
                    "SURVEY 8(d)'s synthetic inputs, inputs resident in HBM; device_us = the call's kernels (per-launch HIP events), "
                    "wall_us = host time per synchronised call; algorithmic bytes per SURVEY 8(d) over device time against 8 TB/s",
           "rows": rows}
    if out_path:
        json.dump(doc, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
