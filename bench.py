#!/usr/bin/env python3
"""bench.py -- create_proof hot path of zero_g's WNN circuit on MI355X (BASELINE.json metric).

A "step" is one pass of the create_proof operation schedule for
model_28input_256entry_1hash_1bpi (k = 14, extended domain 2^17): every commitment MSM and every
NTT of one proof, phase by phase, with a host round trip after each commitment batch exactly where
the Fiat-Shamir transcript needs the points.  Inputs (SRS tables, witness-shaped columns) are
resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W          (N > 1 via torch.distributed.run)

Prints ONE JSON line (rank 0).  value = proofs/hour over all ranks; N > 1 runs independent proof
replicas per GPU (weak scaling, no data-path collective: SURVEY.md 8e / DESIGN.md).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))

import numpy as np
import torch

import zg_halo2 as zg

R = zg.FR_MODULUS
MONT = (1 << 256) % R


def limbs(x):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def to_i64(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.view(np.int64))


def uniform_fr(gen: np.random.Generator, shape) -> np.ndarray:
    """253-bit random limbs: every value is < r, i.e. a valid Montgomery-form element."""
    a = gen.integers(0, 1 << 63, size=tuple(shape) + (4,), dtype=np.int64).astype(np.uint64)
    a[..., :3] ^= gen.integers(0, 1 << 63, size=tuple(shape) + (3,), dtype=np.int64).astype(np.uint64) << np.uint64(1)
    a[..., 3] &= np.uint64((1 << 61) - 1)
    return a


def advice_like_fr(gen: np.random.Generator, shape) -> np.ndarray:
    """Witness-shaped scalars: 70% zero, 20% in {0,1}, 8% bytes, 2% uniform."""
    small = np.array([limbs(v * MONT % R) for v in range(256)], dtype=np.uint64)
    cls = gen.integers(0, 100, size=shape)
    val = np.zeros(tuple(shape), dtype=np.int64)
    m = (cls >= 70) & (cls < 90)
    val[m] = gen.integers(0, 2, size=int(m.sum()))
    m = (cls >= 90) & (cls < 98)
    val[m] = gen.integers(0, 256, size=int(m.sum()))
    out = small[val]
    m = cls >= 98
    out[m] = uniform_fr(gen, (int(m.sum()),))
    return out


class ProofSchedule:
    """Device-resident state + the per-proof launch schedule (appendix B of SURVEY.md)."""

    # model_28input_256entry_1hash_1bpi: k = 14, cs.degree() = 6 -> extended_k = 17, 5 h pieces
    def __init__(self, ctx: zg.Ctx, dev: torch.device, k: int = 14, seed: int = 0):
        self.ctx, self.dev, self.k = ctx, dev, k
        self.n = 1 << k
        self.ext_k = k + 3
        self.en = 1 << self.ext_k
        n, en = self.n, self.en
        gen = np.random.default_rng(seed)
        # SRS (ParamsKZG::new(k)) generated on the GPU, then the window tables of both base sets
        self.d_g = torch.empty((n, 8), dtype=torch.int64, device=dev)
        self.d_gl = torch.empty((n, 8), dtype=torch.int64, device=dev)
        s = np.array(limbs(0x5EED5EED5EED5EED * MONT % R), dtype=np.uint64)
        ctx.params_new_dev(k, s, self.d_g.data_ptr(), self.d_gl.data_ptr())
        ctx.sync()
        self.g = ctx.register_bases_dev(self.d_g.data_ptr(), n)
        self.gl = ctx.register_bases_dev(self.d_gl.data_ptr(), n)
        # witness-shaped columns (Lagrange basis): 6 advice + 1 instance, 8 permuted lookup columns,
        # 2 permutation z + 4 lookup z, 1 random poly, h (extended), 4 GWC witness polys
        self.advice = to_i64(advice_like_fr(gen, (7, n))).to(dev)
        self.permuted = to_i64(uniform_fr(gen, (8, n))).to(dev)
        self.zs = to_i64(uniform_fr(gen, (6, n))).to(dev)
        self.random_poly = to_i64(uniform_fr(gen, (1, n))).to(dev)
        self.h_ext = to_i64(uniform_fr(gen, (1, en))).to(dev)
        self.gwc = to_i64(uniform_fr(gen, (4, n))).to(dev)
        # outputs
        self.ext = torch.empty((21, en, 4), dtype=torch.int64, device=dev)
        self.h_coeff = torch.empty((5 * n, 4), dtype=torch.int64, device=dev)
        self.xyzz = torch.empty((8, 16), dtype=torch.int64, device=dev)
        self.work = torch.empty((21, n, 4), dtype=torch.int64, device=dev)
        self.omega, self.omega_inv = zg.domain_omega(k)
        self.ifft_div = np.array(limbs(pow(n, -1, R) * MONT % R), dtype=np.uint64)
        torch.cuda.synchronize(dev)

    def _commit(self, bases, col: torch.Tensor, batch: int):
        n = self.n
        self.ctx.msm_batch_dev(bases, col.data_ptr(), n, batch, n, self.xyzz.data_ptr())
        return self.ctx.msm_finish(self.xyzz.data_ptr(), batch)  # D2H + normalise: transcript input

    def _intt(self, col: torch.Tensor, slot: int, batch: int):
        # lagrange_to_coeff on a copy (the Lagrange values stay, as in halo2)
        w = self.work[slot:slot + batch]
        w.copy_(col[:batch].view(batch, self.n, 4), non_blocking=True)
        return w

    def step(self):
        ctx, n, k, ek, en = self.ctx, self.n, self.k, self.ext_k, self.en
        pts = []
        # -- advice: 6 commitments (Lagrange basis), 6 + 1 (instance) iNTT
        pts.append(self._commit(self.gl, self.advice, 6))
        w = self.work
        w[0:7].copy_(self.advice.view(7, n, 4))
        w[7:15].copy_(self.permuted.view(8, n, 4))
        w[15:21].copy_(self.zs.view(6, n, 4))
        torch.cuda.current_stream(self.dev).synchronize()
        ctx.ntt_batch_dev(w[0:7].data_ptr(), n, 7, k, self.omega_inv, self.ifft_div)
        # -- theta; lookups commit_permuted: 8 commitments, 8 iNTT
        pts.append(self._commit(self.gl, self.permuted, 8))
        ctx.ntt_batch_dev(w[7:15].data_ptr(), n, 8, k, self.omega_inv, self.ifft_div)
        # -- beta, gamma; permutation (2) + lookup (4) grand products: 6 commitments, 6 iNTT, 2 ext NTT
        pts.append(self._commit(self.gl, self.zs, 6))
        ctx.ntt_batch_dev(w[15:21].data_ptr(), n, 6, k, self.omega_inv, self.ifft_div)
        ctx.coeff_to_extended_batch_dev(w[15:17].data_ptr(), n, self.ext[0:2].data_ptr(), en, 2, k, ek)
        # -- vanishing random polynomial: 1 commitment (coefficient basis)
        pts.append(self._commit(self.g, self.random_poly, 1))
        # -- y; evaluate_h: 7 + 12 coset NTTs (advice+instance, lookup z/a'/s'), then h: 1 ext iNTT
        ctx.coeff_to_extended_batch_dev(w[0:7].data_ptr(), n, self.ext[2:9].data_ptr(), en, 7, k, ek)
        ctx.coeff_to_extended_batch_dev(w[7:19].data_ptr(), n, self.ext[9:21].data_ptr(), en, 12, k, ek)
        ctx.extended_to_coeff_dev(self.h_ext.data_ptr(), k, ek, 5 * n, self.h_coeff.data_ptr())
        pts.append(self._commit(self.g, self.h_coeff, 5))
        # -- x; GWC multiopen: 4 witness commitments
        pts.append(self._commit(self.g, self.gwc, 4))
        return pts


def algorithmic_bytes_per_proof(k: int) -> float:
    n, en = 1 << k, 1 << (k + 3)
    msm = 30 * (n * 96 + 96)
    intt = 21 * 2 * n * 32
    ext = 21 * (n + en) * 32
    ext_inv = (en + 5 * n) * 32
    return float(msm + intt + ext + ext_inv)


def cpu_baseline(k: int, threads: int):
    """The oracle (CPU restatement of halo2's algorithms) timed on this box's host cores over the
    same schedule: 30 MSM (best_multiexp, c = ceil(ln n) per thread chunk), 21 iNTT, 21 coset NTT,
    1 extended iNTT.  kind = "port": halo2's own Rust prover cannot be built here (no cargo)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc

    orc.load().orc_set_threads(threads)
    n = 1 << k
    prm = orc.params_new(k)
    g, gl = prm.g_np(), prm.g_lagrange_np()
    d = orc.domain(6, k)
    adv = [orc.fill_fr_sparse(10 + i, n) for i in range(6)]
    uni = [orc.fill_fr(20 + i, n) for i in range(24)]
    hext = orc.fill_fr(99, 1 << d.extended_k)
    t0 = time.perf_counter()
    for a in adv:
        orc.msm(a, gl, threads)
    for u in uni[:14]:
        orc.msm(u, gl, threads)
    for u in uni[14:24]:
        orc.msm(u, g, threads)
    coeffs = [orc.lagrange_to_coeff(d, a) for a in (adv + uni[:15])]
    for c in coeffs:
        orc.coeff_to_extended(d, c)
    orc.extended_to_coeff(d, hext)
    dt = time.perf_counter() - t0
    return {
        "value": 3600.0 / dt, "unit": "proofs/hour", "cores": threads, "kind": "port",
        "sample": f"1 full MSM+NTT schedule of one k={k} proof (30 MSM, 21 iNTT, 21 coset NTT, 1 ext iNTT) "
                  f"in {dt:.2f} s with the oracle's OpenMP restatement of halo2 best_multiexp/best_fft",
    }


def host_cores() -> int:
    """Cores this process may actually use (the GPU box gives one GPU a 16-core share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--k", type=int, default=14)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group("nccl", device_id=dev)

    ctx = zg.Ctx(local_rank)
    sched = ProofSchedule(ctx, dev, k=args.k, seed=rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        ctx.sync()

    for _ in range(args.warmup):
        sched.step()
    ctx.profile(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sched.step()
    barrier()
    dt = time.perf_counter() - t0
    stats = ctx.profile_collect()
    ctx.profile(False)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        proofs_per_hour = world * args.steps / dt * 3600.0
        # dominant kernel by device time, its algorithmic bytes per launch / its average duration
        dom = max(stats.items(), key=lambda kv: kv[1][1])
        name, (launches, total_ms, abytes) = dom
        avg_ms = total_ms / max(launches, 1)
        achieved = (abytes / max(launches, 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        roofline = {
            "bound": "hbm", "kernel": name, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
            "frac": achieved / 8000.0, "traffic": None,
            "avg_launch_ms": avg_ms, "launches": launches,
            "note": "integer-ALU bound (254-bit Montgomery products); HBM figure reported as BASELINE asks",
        }
        out = {
            "metric": "create_proof proofs/hour (MSM+NTT schedule), model_28input_256entry_1hash_1bpi",
            "value": proofs_per_hour, "unit": "proofs/hour", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32x8 (254-bit Montgomery integers)",
            "data": "synthetic",
            "config": {"workload": f"model_28input_256entry_1hash_1bpi k={args.k}: 30 MSM(2^{args.k}) + 21 iNTT + "
                                   f"21 coset NTT(2^{args.k + 3}) + 1 ext iNTT per proof, 6 transcript round trips",
                       "parallelism": f"{world} proof replica(s), one per GPU"},
            "create_proof_wall_s": ms_per_step / 1e3,
            "algorithmic_GBps": algorithmic_bytes_per_proof(args.k) / (ms_per_step * 1e-3) / 1e9,
            "roofline": roofline,
            "kernels_ms_per_step": {k_: v[1] / args.steps for k_, v in sorted(stats.items(), key=lambda kv: -kv[1][1])},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.k, host_cores())
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
