#!/usr/bin/env python3
"""bench.py -- create_proof of zero_g's WNN circuit on MI355X (BASELINE.json metric).

A "step" is one batch of `--streams` (default 16) full create_proofs, one per proof stream of the GPU,
all in flight together; K steps = K x streams proofs, and value = proofs / hour.  One create_proof
(zg_prover_prove_dev) goes from the assigned advice columns, resident in HBM, to the proof bytes -- 30
commitment MSMs, 21 iNTT + 21 coset NTT + 1 extended iNTT, the 4 lookup arguments (compression,
permutation, grand products), the 2-set permutation argument, evaluate_h over the 2^17-point extended
coset, 67 polynomial evaluations, the 4 GWC openings and the Keccak-256 EvmTranscript -- for zero_g's
WnnCircuit of model_28input_256entry_1hash_1bpi (k = 14) on benches/example_image_7.png: the real
constraint system and the real inference witness (0g-halo2_amd/wnn_circuit.py restates WnnChip; the
class scores it proves are the reference's snapshot, tests/test_wnn_circuit.py).  The SRS tables, the
proving key (fixed / sigma polynomials and cosets) and the witness are in HBM before the timed region,
as in the reference's own bench (benches/bench.rs:30-36 times only `wnn.proof`).

    python bench.py --gpus N --steps K --warmup W          (N > 1 via torch.distributed.run)

Prints ONE JSON line (rank 0).  value = proofs/hour over all ranks.  N > 1 runs independent proofs
per GPU (weak scaling, no data-path collective: proofs do not shard below the MSM, and a 2^14-point
MSM is too small to split -- SURVEY.md 8e / DESIGN.md); the sharded-MSM path is exercised by
tests/test_multi_gpu.py.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
# ROCm multiplexes a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); the
# proof streams (main + side stream each) need their own queues or they serialise behind each other.
# Must be set before the HIP runtime initialises (i.e. before torch touches the GPU).
# 16 proof streams -> 16 queues; RCCL (N > 1: barrier + max over ranks) brings streams of its own, and
# with them 20 queues measure best (tools/sweep_dist.sh: 16 -> 2.06, 20 -> 1.77, 24 -> 1.90 ms/proof).
_dist = int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("ZG_BENCH_FORCE_DIST") == "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20" if _dist else "16")

import numpy as np
import torch

import wnn_circuit
import wnn_model
import zg_halo2 as zg

R = zg.FR_MODULUS
MONT = (1 << 256) % R
# the four configurations of BASELINE.json: (k, model); "large" is a seeded stand-in of the same shape
# because model_49input_8192entry_4hash_6bpi.hdf5 is not in the reference checkout (.MISSING_LARGE_BLOBS)
MODELS = {"tiny": wnn_model.MNIST_TINY, "small": wnn_model.MNIST_SMALL, "medium": wnn_model.MNIST_MEDIUM,
          "large": wnn_model.MNIST_LARGE}


def limbs(x):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def host_cores() -> int:
    """Cores this process may actually use (the GPU box gives one GPU a 16-core share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


class Circuit:
    """Host-side material shared by every proof stream: circuit image, pk values, witness, SRS."""

    def __init__(self, ctx: zg.Ctx, model: str):
        self.k, self.model_name = MODELS[model]
        wnn = wnn_model.synthetic_wnn() if model == "large" else wnn_model.load_checked_in(self.model_name)
        # zero_g's WnnCircuit for this model, synthesised on benches/example_image_7.png: the real
        # constraint system, fixed / sigma columns and witness (wnn_circuit.py restates WnnChip)
        self.cs, self.asg, self.ilen, self.scores = wnn_circuit.build(
            wnn, wnn_model.load_test_image(), self.k, compress_selectors=os.environ.get("ZG_BENCH_NO_SELECTOR_COMPRESSION") != "1")
        k = self.k
        self.img = self.cs.to_c()
        self.fixed, self.sigma = self.asg.fixed_values(), self.asg.sigma_values()
        self.advice = self.asg.advice_values()
        self.instance = self.asg.instance_values(self.ilen)
        self.vk_repr = np.array(limbs(0xC0FFEE * MONT % R), dtype=np.uint64)
        self.s = np.array(limbs(0x5EED5EED5EED5EED * MONT % R), dtype=np.uint64)
        self.g, self.gl = ctx.params_new(k, self.s)  # ParamsKZG::new(k) on the GPU
        # one read-only copy of the MSM window tables per device, shared by every proof stream
        self.g_bases = ctx.register_bases(self.g)
        self.gl_bases = ctx.register_bases(self.gl)


class ProofJob:
    """One proof stream: its own context (HIP stream + workspace) and prover, pk + witness resident
    in HBM; step() = one create_proof."""

    def __init__(self, ctx: zg.Ctx, dev: torch.device, c: Circuit, stream_id: int):
        self.ctx, self.c, self.k, self.cs = ctx, c, c.k, c.cs
        self.prover = zg.Prover(ctx, c.img, c.fixed, c.sigma, c.g_bases, c.gl_bases, c.vk_repr)
        self.prover.set_overlap(False)  # throughput configuration: one HIP stream per proof
        self.d_advice = torch.from_numpy(c.advice.view(np.int64)).to(dev)
        torch.cuda.synchronize(dev)
        self.instance = c.instance
        self.seed = 1000 * stream_id
        self.last = b""

    def step(self):
        # the last blinding_factors+1 rows of every advice column are rewritten by each proof, the
        # other rows are only read: the buffer can be proved from again without a fresh copy
        self.seed += 1
        self.last = self.prover.prove_dev(self.d_advice.data_ptr(), self.instance, self.seed)
        return self.last


def algorithmic_bytes_per_proof(cs) -> float:
    """MSM n*(32+64)+96, NTT 2*n*32, coeff->ext (n+8n)*32, ext->coeff 2*8n*32,
    grand product 3*n*32, evaluate_h (inputs+1)*8n*32."""
    n, en = 1 << cs.k, 1 << cs.extended_k()
    sets = (len(cs.perm_columns) + cs.degree() - 3) // (cs.degree() - 2)
    nl = len(cs.lookups)
    msm = (cs.n_advice + 3 * nl + sets + 1 + (cs.degree() - 1) + 4) * (n * 96 + 96)
    polys = cs.n_advice + cs.n_instance + 3 * nl + sets
    intt = polys * 2 * n * 32
    ext = polys * (n + en) * 32
    ext_inv = 2 * en * 32
    gp = (nl + sets) * 3 * n * 32
    eh_inputs = cs.n_advice + cs.n_instance + cs.n_fixed + len(cs.perm_columns) + sets + 3 * nl + 3
    eh = (eh_inputs + 1) * en * 32
    return float(msm + intt + ext + ext_inv + gp + eh)


def valu_utilisation(ms_per_proof: float):
    """VALU issue utilisation: wave-instructions one proof issues (rocprofv3 --pmc SQ_INSTS_VALU, committed
    under profiles/) per second, against the chip's issue rate of one wave-instruction per SIMD every 4
    cycles (256 CUs x 4 SIMDs x 2.4 GHz / 4).  The multiply-adds that dominate (v_mad_u64_u32 /
    v_mad_i64_i32) occupy the pipe ~1.5x longer than that, so the true pipe occupancy is higher."""
    try:
        v = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")))["valu"]
        per_proof = float(v["wave_instructions_per_proof"])
    except (OSError, KeyError, ValueError):
        return None
    peak = 256 * 4 * 2.4e9 / 4
    # tools/microbench.hip on this chip: a full-rate VALU op (v_add_co_u32) sustains 3.13e13 lane-ops/s =
    # 4.9e11 wave-instructions/s (the clock under load is ~1.9 GHz, not 2.4); v_mad_u64_u32 / v_mad_i64_i32
    # run at 0.87x that
    measured_peak = 31263.2e9 / 64
    achieved = per_proof / (ms_per_proof * 1e-3)
    return {"wave_instructions_per_proof": per_proof, "achieved_wave_instr_per_s": achieved,
            "issue_peak_wave_instr_per_s": peak, "frac": achieved / peak,
            "measured_issue_peak_wave_instr_per_s": measured_peak, "frac_of_measured_peak": achieved / measured_peak}


def cpu_baseline(job: ProofJob, threads: int):
    """The oracle's create_proof (CPU restatement of halo2's algorithms, OpenMP over MSM chunks, FFT
    butterflies and row loops) timed on this box's host cores on the SAME circuit and witness (its own
    seeded SRS of the same size).  kind = "port": halo2's own Rust prover cannot be built here (no cargo/rustc)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc

    orc.load().orc_set_threads(threads)
    c = job.c
    params = orc.params_new(job.k, 0x5EED)
    pk = orc.ProvingKey(c.img, c.fixed, c.sigma, params, c.vk_repr)
    t0 = time.perf_counter()
    st, proof, _ = orc.create_proof(pk, c.advice, c.instance, 1)
    dt = time.perf_counter() - t0
    assert st == 0 and len(proof) == len(job.last)
    return {
        "value": 3600.0 / dt, "unit": "proofs/hour", "cores": threads, "kind": "port",
        "sample": f"1 full create_proof of the same k={job.k} circuit in {dt:.2f} s: oracle/prover.c "
                  f"(plain-C restatement of halo2 create_proof, OpenMP {threads} threads)",
        "wall_s": dt,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", choices=sorted(MODELS), default="tiny",
                    help="tiny = model_28input_256entry_1hash_1bpi (k=14, the BASELINE metric's configuration)")
    ap.add_argument("--streams", type=int, default=16,
                    help="proofs per step = independent proofs in flight per GPU (each on its own HIP stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="no HIP events on the dominant kernel's launches in the timed region (no roofline object)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or os.environ.get("ZG_BENCH_FORCE_DIST") == "1":  # (the env knob rehearses the RCCL path on one GPU)
        import torch.distributed as dist_mod

        dist = dist_mod
        # RCCL announces itself on STDOUT when its communicator comes up ("RCCL version : ..."); the contract is
        # ONE json line there, so file descriptor 1 points at stderr until the first collective has run
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    import threading

    nstreams = max(1, args.streams)
    ctxs = [zg.Ctx(local_rank) for _ in range(nstreams)]
    circuit = Circuit(ctxs[0], args.model)
    jobs = [ProofJob(ctxs[i], dev, circuit, rank * 64 + i) for i in range(nstreams)]
    job = jobs[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        for c in ctxs:
            c.sync()

    def run(steps):
        """`steps` batches: every proof stream proves `steps` witnesses back to back, one host thread
        per stream (ctypes drops the GIL inside the library; streams do not wait for each other)."""

        def work(j, cnt):
            for _ in range(cnt):
                j.step()

        th = [threading.Thread(target=work, args=(jobs[i], steps)) for i in range(nstreams)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    run(max(args.warmup, 1))
    # single-proof latency: one stream alone (transforms overlapped on its side stream), a few proofs;
    # with every launch timed by HIP events once, to learn the per-kernel split and which dominates
    job.prover.set_overlap(True)
    for _ in range(2):
        job.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(3):
        job.step()
    latency_s = (time.perf_counter() - t0) / 3
    phases = job.prover.phase_ms()
    ctxs[0].profile(True)
    for _ in range(2):
        job.step()
    split = ctxs[0].profile_collect()
    ctxs[0].profile(False)
    # the kernel the roofline object describes: the bucket accumulation of the MSM (most field products
    # and most algorithmic bytes of a proof; the top kernel of the rocprofv3 summaries under load).  A lone
    # proof's device-time split can put a latency-bound reduction kernel (a few workgroups) level with
    # it, so the choice is pinned unless another kernel clearly exceeds it.
    top = max(split.items(), key=lambda kv: kv[1][1])[0]
    dominant = "msm_accumulate"
    if dominant not in split or split[top][1] > 1.5 * split[dominant][1]:
        dominant = top
    job.prover.set_overlap(False)  # throughput configuration: one HIP stream per proof
    # timed region: only the dominant kernel's dispatches carry a start / stop event (on the proof's own
    # stream, hipExtLaunchKernelGGL); timing all ~150 launches of a proof costs ~0.7 ms of host time per proof
    for c in ctxs:
        c.profile_filter(dominant)
        c.profile(not args.no_kernel_events)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    stats = {}
    for c in ctxs:
        for name, (l, ms, by) in c.profile_collect().items():
            a = stats.get(name, (0, 0.0, 0.0))
            stats[name] = (a[0] + l, a[1] + ms, a[2] + by)
        c.profile(False)
        c.profile_filter(None)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        ms_per_proof = ms_per_step / nstreams
        proofs_per_hour = world * args.steps * nstreams / dt * 3600.0
        # dominant kernel by device time: its algorithmic bytes per launch / its average duration
        if not stats:
            stats = {"(kernel events disabled)": (1, 0.0, 0.0)}
        name, (launches, total_ms, abytes) = max(stats.items(), key=lambda kv: kv[1][1])
        avg_ms = total_ms / max(launches, 1)
        achieved = (abytes / max(launches, 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        try:  # HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/, separate runs)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")))["kernels"]
            if args.model == "tiny" and name in pmc:
                traffic = pmc[name]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        roofline = {
            "bound": "hbm", "kernel": name, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
            "frac": achieved / 8000.0, "traffic": traffic, "avg_launch_ms": avg_ms,
            "launches_per_proof": launches / (args.steps * nstreams),
            "note": "BASELINE asks for the HBM roofline; the kernels are integer-ALU bound "
                    "(254-bit Montgomery products), see DESIGN.md",
        }
        kernel_ms = sum(v[1] for v in split.values()) / 2
        out = {
            "metric": f"create_proof proofs/hour, {circuit.model_name}",
            "value": proofs_per_hour, "unit": "proofs/hour", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32x8 (254-bit Montgomery integers)",
            "data": ("checked-in model + benches/example_image_7.png (fixtures from the reference checkout), "
                     "seeded SRS" if args.model != "large" else "synthetic (seeded stand-in model), seeded SRS"),
            "config": {"workload": f"full create_proof of zero_g's WnnCircuit for {circuit.model_name} on "
                                   f"example_image_7.png ({job.cs.n_advice} advice, {job.cs.n_fixed} fixed, {len(job.cs.gates)} gates, "
                                   f"{len(job.cs.lookups)} lookups, {len(job.cs.perm_columns)} equality columns, degree {job.cs.degree()}), "
                                   f"k={circuit.k}, EvaluationDomain's extended domain 2^{job.cs.extended_k()} (the throughput form "
                                   f"takes the same quotient from {job.cs.degree() - 1}n points on two cosets), proof {len(job.last)} B"
                                   + (" [seeded stand-in model: the file is absent from the reference]"
                                      if args.model == "large" else ""),
                       "class_scores": circuit.scores,
                       "proofs_per_step": nstreams,
                       "parallelism": f"{world} GPU(s) x {nstreams} independent proof stream(s) per GPU"},
            "proofs_per_step": nstreams, "ms_per_proof": ms_per_proof,
            "create_proof_wall_s": latency_s,
            "streams_per_gpu": nstreams,
            "algorithmic_GBps": algorithmic_bytes_per_proof(job.cs) / (ms_per_proof * 1e-3) / 1e9,
            "single_proof_gpu_kernel_ms": kernel_ms,
            "roofline": roofline,
            # the kernels are VALU-bound: measured instruction count per proof against the issue rate
            "valu": valu_utilisation(ms_per_proof) if args.model == "tiny" else None,
            "single_proof_kernels_ms": {k_: round(v[1] / 2, 4)
                                        for k_, v in sorted(split.items(), key=lambda kv: -kv[1][1])},
            "single_proof_phase_ms": dict(zip(["advice", "lookups_permuted", "products", "h", "evals", "gwc",
                                               "total", "host_sort"], [round(x, 3) for x in phases])),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(job, host_cores())
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
