#!/usr/bin/env python3
"""bench.py -- create_proof of zero_g's WNN circuit on MI355X (BASELINE.json metric).

A "step" is `--provers` (default 12) lock-step batches of `--batch` (default 32) full create_proofs each: every
prover works on its own HIP stream from its own host thread (while one batch waits for its transcript hashes on
the host, the other keeps the GPU busy) and makes its B proofs with ONE launch sequence (zg_prover_prove_batch: the
commitments of a phase are one MSM over B x columns vectors, evaluate_h one grid with a row of workgroups per
proof, ...).  K steps = K x provers x batch proofs, and value = proofs / hour.  One create_proof goes from the
assigned advice columns, resident in HBM, to the proof bytes -- 30 commitment MSMs, 21 iNTT + 21 coset NTT + 1
extended iNTT, the 4 lookup arguments (compression, permutation, grand products), the 2-set permutation argument,
evaluate_h over the extended domain, 67 polynomial evaluations, the 4 GWC openings and the Keccak-256
EvmTranscript -- for zero_g's WnnCircuit of model_28input_256entry_1hash_1bpi (k = 14) on
benches/example_image_7.png: the real constraint system and the real inference witness
(harness/wnn_circuit.py restates WnnChip; the class scores it proves are the reference's snapshot,
tests/test_wnn_circuit.py).  The SRS tables, the proving key (fixed / sigma polynomials and cosets, ONE copy shared
by the provers) and the witness are in HBM before the timed region, as in the reference's own bench
(benches/bench.rs:30-36 times only `wnn.proof`).  What is timed is what is checked: after the timed region the proofs
of the last step are compared byte for byte with the oracle's and pairing-verified ("verified" in the line).

    python bench.py --gpus N --steps K --warmup W [--mode replicas|shard-msm]   (N > 1 via torch.distributed.run)

Prints ONE JSON line (rank 0).  value = proofs/hour over all ranks.
  --mode replicas (default): N > 1 runs independent proofs per GPU (weak scaling, no data-path collective).
  --mode shard-msm: the commitments of every proof are sharded by point range over the N GPUs
      (zg_prover_set_shard: each rank multiplies its slice of ParamsKZG::g / ::g_lagrange, one all-gather of the
      partial sums per commitment phase over RCCL, local EC additions); transforms and evaluate_h stay per GPU, every
      rank ends with the same proof bytes.  Strong scaling of the MSM share of a proof only.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
sys.path.insert(0, os.path.join(ROOT, "harness"))
# ROCm multiplexes a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); every prover stream
# should have a queue of its own, next to torch's and RCCL's.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# Kernel arguments in device memory instead of host memory read over PCIe at every dispatch: -2.4 % on a lone proof's 73
# launches (2.17 / 2.20 / 2.21 -> 2.14 / 2.15 / 2.13 ms, three alternations on one box, profiles/r04/ab_runtime_env.txt),
# nothing on the throughput form.  A setting of the HIP runtime, read when it initialises: the process's to make
# (INTEGRATION.md), not the library's.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# Completion signals polled instead of waited for through an interrupt: a lone proof's host waits six times for the device
# (2.121 / 2.130 / 2.133 -> 2.097 / 2.119 / 2.090 ms, three alternations, profiles/r04/ab_runtime_env.txt); the throughput
# form is unchanged (0.6601 against 0.6606 ms/proof) -- its twelve host threads wait in turns.
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
RUNTIME_ENV = {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG", "HSA_ENABLE_INTERRUPT")}

import numpy as np
import torch

import wnn_circuit
import wnn_model
import zg_halo2 as zg

R = zg.FR_MODULUS
MONT = (1 << 256) % R
PROFILES = os.path.join(ROOT, "profiles", os.environ.get("ZG_BENCH_PROFILES", "r04"))  # the counter files the line reads
# the four configurations of BASELINE.json: (k, model); "large" is a seeded stand-in of the same shape
# because model_49input_8192entry_4hash_6bpi.hdf5 is not in the reference checkout (.MISSING_LARGE_BLOBS)
MODELS = {"tiny": wnn_model.MNIST_TINY, "small": wnn_model.MNIST_SMALL, "medium": wnn_model.MNIST_MEDIUM,
          "large": wnn_model.MNIST_LARGE}
# chip constants (/opt/skills/guides/MI355X_MICROARCH.md): HBM3E peak; a SIMD issues one wave64 VALU instruction
# every 2 cycles (SIMD-32 datapath), 256 CUs x 4 SIMDs at 2.4 GHz
HBM_PEAK_GBPS = 8000.0
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2


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
    """Host-side material of one model: circuit image, pk values, witness, SRS (point range [lo, hi) of it)."""

    def __init__(self, ctx: zg.Ctx, model: str, shard=(0, 1)):
        self.k, self.model_name = MODELS[model]
        self.model = model
        wnn = wnn_model.synthetic_wnn() if model == "large" else wnn_model.load_checked_in(self.model_name)
        self.wnn = wnn
        # zero_g's WnnCircuit for this model, synthesised on benches/example_image_7.png: the real
        # constraint system, fixed / sigma columns and witness (harness/wnn_circuit.py restates WnnChip)
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
        rank, world = shard
        n = 1 << k
        self.lo, self.hi = rank * n // world, (rank + 1) * n // world
        # one read-only copy of the MSM window tables per device, shared by every prover
        self.g_bases = ctx.register_bases(self.g[self.lo:self.hi])
        self.gl_bases = ctx.register_bases(self.gl[self.lo:self.hi])


HOST_ADVICE = False  # --host-advice: measure the PCIe-inclusive rate (never the headline `value`)


class Stream:
    """One proof stream: a prover (own context = own HIP stream + workspace; proving key shared with the others) with
    `batch` slots whose advice columns are resident in HBM; step() = one lock-step batch of create_proofs."""

    def __init__(self, ctx: zg.Ctx, prover: zg.Prover, c: Circuit, batch: int, stream_id: int, exchange=None, shard=(0, 1)):
        self.ctx, self.prover, self.c, self.batch = ctx, prover, c, batch
        prover.set_batch(batch)
        prover.set_overlap(False)  # throughput configuration: one HIP stream per prover, split extended domain
        # `exchange` is THIS prover's: a communicator serialises its collectives on one stream, so every prover of a
        # rank has one of its own (an RcclComm: the all-gather runs inside the library; else a host callback)
        if shard[1] > 1 and hasattr(exchange, "handle"):
            prover.set_shard_rccl(shard[0], shard[1], c.lo, exchange.handle)
        elif shard[1] > 1 and hasattr(exchange, "c_fn"):  # a C function pointer (tools/shard_compute_leg.py's stub)
            prover.set_shard_c(shard[0], shard[1], c.lo, exchange.c_fn, exchange.c_user)
        elif shard[1] > 1:
            prover.set_shard(shard[0], shard[1], c.lo, exchange)
        # the witness into every slot, once: a proof rewrites only the last blinding_factors+1 rows of its advice
        # columns and reads the rest, so the slots can be proved from again (inputs resident in HBM, as the contract asks)
        self.seed0 = 1_000_000 * stream_id
        self.stream_id = stream_id
        self.steps = 0
        self.plan = None
        self.last = self.prover.prove_batch([c.advice] * batch, [c.instance] * batch, self.seeds())[0]

    def enable_images(self, arrays: dict, pool: np.ndarray):
        """From here on every proof of a step is for ANOTHER image of `pool`: the recorded witness program runs on the
        device (zg_witness_run_dev) into the prover's slots -- image bytes in, class scores out, inside step()."""
        self.plan = zg.WitnessPlan(self.ctx, arrays)
        self.pool = pool
        self.slots = [self.prover.advice_slot(b) for b in range(self.batch)]

    def seeds(self):
        return [self.seed0 + 1000 * self.steps + b for b in range(self.batch)]

    def step(self):
        self.steps += 1
        self.last_seeds = self.seeds()
        if self.plan is not None:
            first = (self.stream_id * 7 + self.steps * self.batch) % len(self.pool)
            self.last_images = [(first + b) % len(self.pool) for b in range(self.batch)]
            self.last_inst = self.plan.run(self.pool[self.last_images], self.slots)
            self.last = self.prover.prove_batch(None, [i[None, :, :] for i in self.last_inst], self.last_seeds, device=True)[0]
        elif HOST_ADVICE:  # the host-pointer entry: every proof's columns cross PCIe (3 MiB per proof at k = 14)
            self.last = self.prover.prove_batch([self.c.advice] * self.batch, [self.c.instance] * self.batch, self.last_seeds)[0]
        else:
            self.last = self.prover.prove_batch(None, [self.c.instance] * self.batch, self.last_seeds, device=True)[0]
        return self.last


def make_streams(dev_index: int, c: Circuit, ctx0: zg.Ctx, nprovers: int, batch: int, rank: int, exchanges=None, shard=(0, 1),
                 probe=None):
    """The provers of one GPU: the first, then its forks (same proving key and base tables) on contexts of their own.
    ORDER MATTERS: HIP hands out hardware queues in stream-creation order and the chip runs four compute pipes, so
    streams whose queue indices are equal mod 4 share a pipe.  The four throughput streams are therefore created first
    and back to back (queues 0-3: one pipe each; any other order measured 0.89 instead of 0.83 ms/proof), and the lone
    proof of the latency probe runs on a further fork created after them, whose main and side stream land on queues 4
    and 5 -- two different pipes again (3.0 ms; 3.8 ms when the pair shares one)."""
    base = 0 if shard[1] > 1 else rank * 64  # (ranks of a sharded proof prove the SAME statements with the same keys)
    ctxs = [ctx0] + [zg.Ctx(dev_index) for _ in range(nprovers - 1)]
    first = zg.Prover(ctx0, c.img, c.fixed, c.sigma, c.g_bases, c.gl_bases, c.vk_repr)
    first.set_overlap(False)  # (before forking: a fork of a single-stream prover creates no side stream of its own)
    provers = [first] + [first.fork(x) for x in ctxs[1:]]
    streams = [Stream(ctxs[i], provers[i], c, batch, base + i, exchanges[i] if exchanges else None, shard) for i in range(nprovers)]
    probed = None
    if probe:
        pctx = zg.Ctx(dev_index)
        lone = Stream(pctx, first.fork(pctx), c, 1, base + 63)
        probed = probe(lone)
        lone.prover.close()
        pctx.close()
    return ctxs, streams, probed


def run_steps(streams, steps):
    """`steps` batches per stream, one host thread per stream (ctypes drops the GIL inside the library)."""
    errors = []

    def work(s):
        try:
            for _ in range(steps):
                s.step()
                if WATCHDOG:
                    WATCHDOG.pat()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=work, args=(s,)) for s in streams]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        raise errors[0]


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


FAMILIES = {"msm": ("msm_",), "ntt": ("ntt_",), "evaluate_h": ("evaluate_h",), "sort": ("sort_", "permute_"),
            "products": ("grand_product", "lookup_", "perm_terms", "permuted_finish", "blind_rows", "random_poly"),
            "openings": ("eval_dot", "powers", "horner_combine", "kate_", "fold", "diff_scale", "split_combine")}  # (ZG_LAUNCH labels)


def family_of(kernel: str) -> str:
    for fam, prefixes in FAMILIES.items():
        if kernel.startswith(prefixes):
            return fam
    return "other"


def load_pmc():
    """Counter figures of the same configuration (rocprofv3 --pmc passes, tools/profile.sh + tools/install_profile.py;
    committed under profiles/rNN): HBM bytes per launch per kernel and VALU wave-instructions per proof."""
    return load_json("pmc_traffic.json")


def load_json(name: str):
    """a counter file of this round's profile set, else of the newest earlier round that has it (`_from` says which)"""
    rounds = [PROFILES] + sorted((os.path.join(ROOT, "profiles", d) for d in os.listdir(os.path.join(ROOT, "profiles"))
                                  if d.startswith("r") and os.path.join(ROOT, "profiles", d) < PROFILES), reverse=True)
    for d in rounds:
        try:
            out = json.load(open(os.path.join(d, name)))
            out["_from"] = os.path.relpath(os.path.join(d, name), ROOT)
            return out
        except (OSError, ValueError):
            continue
    return None


# kernel families whose unit of work SURVEY.md 8d defines (its per-proof figure is their sum); the others are charged what
# their kernels stream
UNIT_FAMILIES = ("msm", "ntt", "evaluate_h", "products")


def roofline_tables(stats: dict, pmc, pmc_scale: float):
    """Per kernel and per family from the per-launch HIP events of one measured region.
    stats[kernel] = (launches, device ms, streamed bytes, unit bytes) as the library charges every launch (include/zg_halo2.h,
    zg_kernel_stat): `streamed` = what the kernel's own algorithm moves (each distinct input once, each output once), `unit` =
    SURVEY.md 8d's figure for the unit of work, charged ONCE per unit on the kernel that carries it (an MSM's n * 96 + 96 on
    msm_accumulate, not on each of its eight kernels).  A kernel's GB/s is its own streamed bytes over its own time; a family's
    is its units' bytes (where SURVEY defines the unit, else its kernels' streams) over the family's time, and its counter
    traffic is the SUM over its kernels -- the MSM's eight kernels together against ONE n * 96 + 96."""
    kernels, fam = {}, {}
    for name, (l, ms, by, ub) in stats.items():
        k_pmc = (pmc or {}).get("kernels", {}).get(name)
        gbps = (by / (ms * 1e-3) / 1e9) if ms > 0 else 0.0
        kernels[name] = {"launches": l, "total_ms": round(ms, 3), "avg_launch_ms": ms / max(l, 1),
                         "algo_bytes_per_launch": by / max(l, 1), "unit_bytes_per_launch": ub / max(l, 1),
                         "algo_GBps": gbps,
                         "hbm_bytes_per_launch": int(k_pmc["hbm_bytes_per_launch"] * pmc_scale) if k_pmc else None}
        f = fam.setdefault(family_of(name), {"total_ms": 0.0, "streamed": 0.0, "unit": 0.0, "hbm_bytes": 0.0, "hbm_known": True})
        f["total_ms"] += ms
        f["streamed"] += by
        f["unit"] += ub
        if k_pmc:
            f["hbm_bytes"] += k_pmc["hbm_bytes_per_launch"] * pmc_scale * l
        else:
            f["hbm_known"] = False
    device_ms = sum(v[1] for v in stats.values())
    families = {}
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["total_ms"]):
        unit_basis = name in UNIT_FAMILIES and f["unit"] > 0
        algo = f["unit"] if unit_basis else f["streamed"]
        gbps = algo / (f["total_ms"] * 1e-3) / 1e9 if f["total_ms"] > 0 else 0.0
        families[name] = {"share_of_device_time": f["total_ms"] / device_ms if device_ms else 0.0,
                          "algorithmic_bytes": algo, "basis": "SURVEY 8d units, each charged once" if unit_basis else "what the kernels stream",
                          "streamed_bytes_of_its_kernels": f["streamed"],
                          "algo_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS,
                          "counter_bytes": f["hbm_bytes"] if f["hbm_known"] else None,
                          "counter_over_algorithmic_bytes": (f["hbm_bytes"] / algo) if f["hbm_known"] and algo else None}
    charged_unit_bytes = sum(f["unit"] for f in fam.values())
    return kernels, families, device_ms, charged_unit_bytes


def cpu_baseline(c: Circuit, proof_len: int, threads: int, repeats: int = 7):
    """The oracle's create_proof (CPU restatement of halo2's algorithms, OpenMP over MSM chunks, FFT butterflies and row
    loops) timed on this box's host cores on the SAME circuit and witness (its own seeded SRS of the same size): the
    MEDIAN of `repeats` proofs after one untimed warm-up, every sample and the per-phase split of
    the median proof in the record.  kind = "port": halo2's own Rust prover cannot be built here (no cargo/rustc)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc

    orc.load().orc_set_threads(threads)
    params = orc.params_new(c.k, 0x5EED)
    pk = orc.ProvingKey(c.img, c.fixed, c.sigma, params, c.vk_repr)
    st, proof, _ = orc.create_proof(pk, c.advice, c.instance, 1)  # warm-up: page in, OpenMP team up
    assert st == 0 and len(proof) == proof_len
    samples, phases = [], []
    for i in range(repeats):
        t0 = time.perf_counter()
        st, proof, _ = orc.create_proof(pk, c.advice, c.instance, 2 + i)
        samples.append(time.perf_counter() - t0)
        phases.append(orc.last_phase_ms())
        assert st == 0 and len(proof) == proof_len
    order = sorted(range(repeats), key=lambda i: samples[i])
    mid = order[repeats // 2]
    dt = samples[mid]
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    return {
        "value": 3600.0 / dt, "unit": "proofs/hour", "cores": threads, "kind": "port",
        "sample": f"median of {repeats} full create_proofs of the same k={c.k} circuit (after 1 warm-up): {dt:.3f} s each; "
                  f"oracle/prover.c (plain-C restatement of halo2 create_proof, OpenMP {threads} threads); its evaluate_h "
                  f"INTERPRETS the gate monomials row by row where halo2's GraphEvaluator runs a compiled graph (the `h` phase is "
                  f"~3/4 of this proof), on {threads} of the machine's {os.cpu_count()} cores: expected SLOWER than real halo2 on the "
                  f"same cores -- a baseline to be read with that, never a speed-up claim",
        "wall_s": dt, "samples_s": [round(x, 4) for x in samples], "min_s": min(samples), "max_s": max(samples),
        "openmp": {"threads": threads, "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"), "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"),
                   "cores_in_affinity_mask": affinity, "machine_cores": os.cpu_count()},
        "phase_ms": dict(zip(["advice", "lookups_permuted", "products", "h", "evals", "gwc", "total"],
                             [round(x, 2) for x in phases[mid][:7]])),
    }


def verify_last_step(c: Circuit, streams, threads: int) -> dict:
    """What was timed is what is checked: the proofs the streams made in the LAST timed step against the oracle --
    byte for byte for the first and the last proof of stream 0's batch and the first of every other stream, the public
    pairing equation for one of them.  (The oracle's SRS is rebuilt from the bench's toxic scalar.)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc

    orc.load().orc_set_threads(threads)
    params = orc.params_from_scalar(c.k, c.s)
    pk = orc.ProvingKey(c.img, c.fixed, c.sigma, params, c.vk_repr)
    checked = 0
    picks = [(0, 0), (0, streams[0].batch - 1)] + [(i, 0) for i in range(1, len(streams))]
    for i, b in dict.fromkeys(picks):
        s = streams[i]
        st, want, _ = orc.create_proof(pk, c.advice, c.instance, s.last_seeds[b])
        if st != 0 or want != s.last[b]:
            return {"verified": False, "detail": f"stream {i} proof {b} differs from the oracle"}
        checked += 1
    pick = streams[-1].last[-1]
    if orc.verify_proof_pairing(pk, c.instance, pick) != 1:
        return {"verified": False, "detail": "pairing check failed"}
    return {"verified": True, "detail": f"{checked} proofs of the last timed step byte-identical to the oracle's, 1 pairing-verified"}


def image_pool(c: Circuit, count: int = 64) -> np.ndarray:
    """benches/example_image_7.png and seeded noise images of its shape (synthetic: the MNIST test set is not in the
    reference checkout)"""
    real = wnn_model.load_test_image()
    rng = np.random.default_rng(2024)
    return np.stack([real] + [rng.integers(0, 256, size=real.shape, dtype=real.dtype) for _ in range(count - 1)]).reshape(count, -1)


def image_to_proof(c: Circuit, streams, ctxs, barrier, threads: int, verify: bool, steps: int = 5, checks: int = 3) -> dict:
    """The same provers, but every proof for a different image and the witness made on the device inside the timed
    step (Wnn::proof's whole body, /root/reference/src/wnn.rs:232-262, image bytes to proof bytes).
    Checked like the headline: proofs of the last step against the oracle's create_proof of the HOST-synthesised witness
    of the same image."""
    import witness_tape

    t0 = time.perf_counter()
    prog = witness_tape.trace(c.wnn, c.k)
    arrays = prog.arrays()
    trace_s = time.perf_counter() - t0
    pool = image_pool(c)
    for s in streams:
        s.enable_images(arrays, pool)
    dt, stats = measure(streams, ctxs, steps, 1, barrier, profile=True)
    n = steps * sum(s.batch for s in streams)
    wit_ms = sum(stats.get(k_, (0, 0.0, 0.0, 0.0))[1] for k_ in ("witness_run", "witness_finish"))
    wit_launches = stats.get("witness_run", (0, 0.0, 0.0, 0.0))[0]
    out = {"ms_per_proof": dt / n * 1e3, "proofs_per_hour": n / dt * 3600.0, "images": len(pool),
           "witness_program": {"operations": int(arrays["ops"].shape[0]), "levels": int(arrays["level_start"].shape[0] - 1),
                               "assigned_cells": len(prog.cells), "recorded_in_s": round(trace_s, 2)},
           "witness_device_ms_per_batch": wit_ms / max(1, wit_launches),
           "note": "every proof of a step for another image; witness program + create_proof inside the timed region; "
                   "per image 784 B in and the class scores out cross PCIe"}
    if verify:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orc

        orc.load().orc_set_threads(threads)
        params = orc.params_from_scalar(c.k, c.s)
        pk = orc.ProvingKey(c.img, c.fixed, c.sigma, params, c.vk_repr)
        ok, checked = True, 0
        shape = wnn_model.load_test_image().shape
        for i, b in list(dict.fromkeys([(0, 0), (0, streams[0].batch - 1), (len(streams) - 1, 0)]))[:checks]:
            s = streams[i]
            im = pool[s.last_images[b]].reshape(shape)
            _, asg, ilen, scores = wnn_circuit.build(c.wnn, im, c.k)
            inst = asg.instance_values(ilen)
            st, want, _ = orc.create_proof(pk, asg.advice_values(), inst, s.last_seeds[b])
            ok = ok and st == 0 and want == s.last[b] and np.array_equal(inst[0], s.last_inst[b]) and scores == c.wnn.predict(im)
            checked += 1
        out["verified"] = bool(ok)
        out["detail"] = f"{checked} proofs of the last step byte-identical to the oracle's proof of the host-synthesised witness of the same image"
    for s in streams:
        s.plan.close()
        s.plan = None
    return out


def measure(streams, ctxs, steps, warmup, barrier, profile=False):
    if WATCHDOG:
        WATCHDOG.arm(True)
    run_steps(streams, max(warmup, 1))
    for x in ctxs:
        x.profile(profile)
    barrier()
    t0 = time.perf_counter()
    run_steps(streams, steps)
    barrier()
    dt = time.perf_counter() - t0
    if WATCHDOG:
        WATCHDOG.arm(False)
    stats = {}
    for x in ctxs:
        if profile:
            for name, (l, ms, by, ub) in x.profile_collect().items():
                a = stats.get(name, (0, 0.0, 0.0, 0.0))
                stats[name] = (a[0] + l, a[1] + ms, a[2] + by, a[3] + ub)
        x.profile(False)
    return dt, stats


class stdout_to_stderr:
    """RCCL and gloo announce themselves on STDOUT when a communicator / group comes up ("RCCL version : ...", "[Gloo] Rank 0
    is connected to ..."); the contract is ONE json line there, so file descriptor 1 points at stderr meanwhile."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def latency_probe(stream: Stream):
    """One proof alone: transforms overlapped on a side stream, several lanes per EC addition (set_overlap(True))."""
    p, c = stream.prover, stream.c
    p.set_overlap(True)
    # the lone-proof digit tables (78 GB at k = 14): an explicit call since round 4, here before the timed proofs
    table_bytes = p.enable_digit_tables()
    # ... and so is the gate (ZG_LAT_GATE, include/zg_halo2.h): each phase queued behind a kernel that waits for the host's
    # challenge.  Opt-in in the library -- it needs every stream of the process on a hardware queue of its own, which this
    # process arranges (GPU_MAX_HW_QUEUES above) and during this probe no other stream has work.  LONE_PROOF_GATE says so
    # in the line; ZG_LAT_GATE=0 in the environment keeps the probe plain.
    gate = os.environ.get("ZG_LAT_GATE", "1") != "0"
    if gate:
        zg.tuning_set("ZG_LAT_GATE", 1)
    try:
        for _ in range(3):
            p.prove_dev(p.advice_slot(0), c.instance, 1)
        each = []
        for i in range(9):  # (one proof per measurement: the median of nine)
            t0 = time.perf_counter()
            p.prove_dev(p.advice_slot(0), c.instance, 2 + i)
            each.append(time.perf_counter() - t0)
        latency_s = sorted(each)[len(each) // 2]
        latency_probe.samples_ms = [round(x * 1e3, 4) for x in each]
    finally:
        if gate:
            zg.tuning_set("ZG_LAT_GATE", -1)
    phases = p.phase_ms()
    p.set_overlap(False)
    latency_probe.gate = gate
    return latency_s, phases, table_bytes


def launcher_command(argv, gpus: int, env) -> "list[str] | None":
    """What `python bench.py --gpus N ...` has to START, decided before anything touches the GPU (the harness being
    mirrored, /root/reference/benches/bench.rs:47-76, is one process per measurement; the contract here is one process
    per GPU).  Returns None when THIS process is a rank (WORLD_SIZE set by torch.distributed.run, or N = 1), else the
    command of the child that brings up N ranks: this process then only relays the child's JSON line and exit code.
    A WORLD_SIZE that contradicts --gpus is an error (a one-GPU number must never be printed as an N-GPU line)."""
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            raise SystemExit(f"bench.py: --gpus {gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {gpus}` "
                             f"(it launches its own ranks) or give torch.distributed.run --nproc-per-node {gpus}")
        return None
    if gpus <= 1:
        return None
    port = env.get("ZG_BENCH_PORT") or str(free_port())
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", port, os.path.abspath(__file__)] + list(argv)


def free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relay(cmd) -> int:
    """Run the ranks as a CHILD process (never os.exec*: this process may not be replaced once a GPU runtime could be
    loaded), pass its stderr through, print the one JSON line of its rank 0 and return its exit code."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (RCCL between processes needs dmabuf IPC on this pool)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in child.stdout:
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln:
            print(ln, file=sys.stderr)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited 0 without a JSON line", file=sys.stderr)
        rc = 4
    return rc


class Watchdog:
    """A stuck collective must end the run non-zero, not hang it: every rank pats the dog when a step of any
    of its provers completes; ZG_BENCH_STALL_S seconds (default 300) without progress inside a timed or warm-up region
    -> stacks to stderr, exit code 3."""

    def __init__(self, seconds: float):
        self.seconds, self.last, self.armed = seconds, time.monotonic(), False
        threading.Thread(target=self._run, daemon=True).start()

    def pat(self):
        self.last = time.monotonic()

    def arm(self, on: bool):
        self.last, self.armed = time.monotonic(), on

    def _run(self):
        import faulthandler

        while True:
            time.sleep(1.0)
            if self.armed and time.monotonic() - self.last > self.seconds:
                print(f"bench.py: no prover finished a step for {self.seconds:.0f} s (stuck collective?): giving up", file=sys.stderr)
                faulthandler.dump_traceback(file=sys.stderr)
                os._exit(3)


WATCHDOG = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", choices=sorted(MODELS), default="tiny",
                    help="tiny = model_28input_256entry_1hash_1bpi (k=14, the BASELINE metric's configuration)")
    ap.add_argument("--batch", type=int, default=32, help="proofs per lock-step batch (zg_prover_prove_batch)")
    ap.add_argument("--provers", type=int, default=None,
                    help="proof streams per GPU (provers sharing one proving key): 12; in shard-msm 4 with the host exchange "
                         "(an exchange group per prover) and 1 with raw RCCL communicators")
    ap.add_argument("--mode", choices=["replicas", "shard-msm"], default="replicas")
    ap.add_argument("--exchange", choices=["host", "rccl"], default=None,
                    help="shard-msm: all-gather inside the library on a raw RCCL communicator per prover "
                         "(zg_prover_set_shard_rccl; the default when every rank owns a GPU: ONE prover per rank then) or "
                         "through a host callback on torch.distributed groups, one per prover (the default of one-GPU "
                         "rehearsals: RCCL refuses two ranks on one device)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="no per-launch HIP events in the timed region (no roofline object)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of the other three models")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--host-advice", action="store_true",
                    help="every proof uploads its advice columns from host memory (PCIe-inclusive rate, for DESIGN.md)")
    ap.add_argument("--no-serialised", action="store_true", help="skip the one-prover pass behind roofline.serialised")
    ap.add_argument("--no-latency-probe", action="store_true", help="skip the lone-proof latency measurement (counter passes)")
    ap.add_argument("--no-image-to-proof", action="store_true", help="skip the run with a different image per proof (device witness)")
    args = ap.parse_args()

    global HOST_ADVICE, WATCHDOG
    HOST_ADVICE = args.host_advice
    # --gpus N > 1 outside torch.distributed.run: start the N ranks as a child and relay its line -- decided BEFORE the
    # first GPU call of this process (torch.cuda.is_available() below initialises the runtime)
    cmd = launcher_command(sys.argv[1:], args.gpus, os.environ)
    if cmd is not None:
        raise SystemExit(relay(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # (rehearsal knobs for a one-GPU box: ZG_BENCH_DEVICE=0 puts every rank on the same card, ZG_BENCH_BACKEND=gloo
    #  replaces RCCL, which refuses two ranks on one GPU)
    dev_index = int(os.environ.get("ZG_BENCH_DEVICE", local_rank))
    backend = os.environ.get("ZG_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    dist, collective_ranks = None, 1
    if world > 1 or os.environ.get("ZG_BENCH_FORCE_DIST") == "1":  # (the env knob rehearses the RCCL path on one GPU)
        import torch.distributed as dist_mod

        dist = dist_mod
        with stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            dist.barrier()
            torch.cuda.synchronize()
            # what the collective backend itself saw: an all-reduce of ones over the default group (RCCL when every rank
            # owns a GPU) -- the N of an N-GPU line is this number, not an argument echoed back
            ones = torch.ones(1, dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(ones)
            collective_ranks = int(ones.item())
        if collective_ranks != world:
            raise SystemExit(f"bench.py: the {backend} group counts {collective_ranks} ranks, WORLD_SIZE says {world}")
        WATCHDOG = Watchdog(float(os.environ.get("ZG_BENCH_STALL_S", "300")))

    sharded = args.mode == "shard-msm" and world > 1
    shard = (rank, world) if sharded else (0, 1)
    # every rank on a GPU of its own (the driver's launch): RCCL; ranks sharing a card (rehearsal): host callback
    own_gpu = backend == "nccl" and "ZG_BENCH_DEVICE" not in os.environ
    exchange_kind = (args.exchange or ("rccl" if own_gpu else "host")) if sharded else None
    # shard-msm over raw RCCL communicators: ONE prover per rank unless --provers says otherwise.  Several provers of a
    # process would issue ncclAllGather on several communicators from several threads, in an order that differs from rank
    # to rank -- the classic NCCL/RCCL deadlock once the collectives' kernels cannot all be resident -- and no
    # world > 1 run has ever validated it; the host exchange (gloo / torch.distributed groups, one per prover) has no
    # device-side spinning and keeps 4.
    nprov = max(1, args.provers if args.provers else (12 if not sharded else 1 if exchange_kind == "rccl" else 4))
    batch = max(1, args.batch)
    if args.model == "large":
        batch = min(batch, 8)  # (a k = 17 proof slot is 1.4 GiB; 8 provers x 8 slots + workspaces stay well inside 288 GB)
    exchanges = None
    # rccl_ranks: the ranks RCCL itself counted -- the all-reduce above when the default group is RCCL (replicas and the
    # host exchange), ncclCommCount of the prover's communicator in shard-msm's in-library exchange; null under gloo
    rccl_ranks = collective_ranks if (dist is not None and backend == "nccl") else None
    if sharded:
        import multi_gpu

        # ONE exchange PER PROVER (prover i of every rank forms a group with prover i of the others): the provers of a
        # rank work through their phases independently, each on its own stream / host thread
        with stdout_to_stderr():
            if exchange_kind == "rccl":
                exchanges = [multi_gpu.RcclComm(rank, world, dev_index, dist) for _ in range(nprov)]
                rccl_ranks = exchanges[0].count()
            else:
                groups = [dist.new_group(backend=None) for _ in range(nprov)]  # (collective calls: same order on every rank)
                exchanges = [multi_gpu.make_exchange(dist, dev if backend == "nccl" else None, g) for g in groups]
                for g in groups:  # (a group's transport comes up at its first collective: now, not inside the timed region)
                    dist.barrier(group=g)

    ctx0 = zg.Ctx(dev_index)
    circuit = Circuit(ctx0, args.model, shard)
    want_probe = not (sharded or args.no_latency_probe)
    ctxs, streams, probed = make_streams(dev_index, circuit, ctx0, nprov, batch, rank, exchanges, shard,
                                         probe=latency_probe if want_probe else None)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        for x in ctxs:
            x.sync()

    latency_s, phases, table_bytes = probed if probed else (None, [0.0] * 8, 0)
    lone_samples_ms, lone_gate = (getattr(latency_probe, "samples_ms", None), getattr(latency_probe, "gate", None)) if probed else (None, None)
    # timed region: every launch carries its own start / stop event (hipExtLaunchKernelGGL on the prover's stream): a
    # lock-step batch is ~100 launches for `batch` proofs, so timing them all costs nothing measurable
    dt, stats = measure(streams, ctxs, args.steps, args.warmup, barrier, profile=not args.no_kernel_events)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        proofs_per_step = nprov * batch
        n_proofs = args.steps * proofs_per_step * (1 if sharded else world)
        ms_per_step = dt / args.steps * 1e3
        ms_per_proof = dt / (args.steps * proofs_per_step) * 1e3
        proofs_per_hour = n_proofs / dt * 3600.0
        pmc = load_pmc() if args.model == "tiny" and not sharded else None
        # (the counter passes were taken at pmc["proofs_per_launch"] proofs per launch: every kernel of the path processes
        #  the proofs of a batch side by side, so bytes per launch scale with the batch)
        pmc_scale = batch / float(pmc.get("proofs_per_launch", batch)) if pmc else 1.0
        launches_per_proof = sum(v[0] for v in stats.values()) / max(1, args.steps * proofs_per_step)
        kernels, families, device_ms, charged_unit_bytes = roofline_tables(stats, pmc, pmc_scale)
        # ONE prover stepping, same lock-step batches, every launch bracketed by its own events -- one stream, so the kernels
        # run one at a time: what a kernel costs ALONE on the chip.  The timed region's durations are shared-chip durations
        # (eleven other provers run beside every launch) and flip from box to box; the dominant kernel is picked HERE.
        serial = None
        if stats and not args.no_serialised and not sharded and world == 1:
            dt1, st1 = measure(streams[:1], ctxs[:1], 3, 1, barrier, profile=True)
            k1, f1, dev1, _ = roofline_tables(st1, pmc, pmc_scale)
            serial = {"note": "one prover alone on the chip (one stream: kernels run one at a time), same batches of "
                              f"{batch}; per-launch HIP events; algorithmic bytes / launch duration against the HBM peak",
                      "ms_per_proof": dt1 / (3 * batch) * 1e3, "device_ms_per_proof": dev1 / (3 * batch),
                      "kernels": {k_: {"avg_launch_ms": round(v["avg_launch_ms"], 4), "share_of_device_time": round(v["total_ms"] / dev1, 4) if dev1 else 0.0,
                                       "algo_GBps": round(v["algo_GBps"], 1), "frac_of_hbm_peak": round(v["algo_GBps"] / HBM_PEAK_GBPS, 5)}
                                  for k_, v in sorted(k1.items(), key=lambda kv: -kv[1]["total_ms"])},
                      "families": f1}
        roofline = None
        if stats:
            # the dominant kernel: the largest share of the SERIALISED pass; without one (N > 1, --no-serialised) the largest
            # share of the proof's VALU instructions (counter file), else of this run's shared-chip time
            by_valu = (pmc or {}).get("valu", {}).get("by_kernel") or {}
            if serial:
                name, picked_by = next(iter(serial["kernels"])), "largest device time in the serialised pass (one prover alone on the chip)"
            elif by_valu:
                name, picked_by = max((k_ for k_ in by_valu if k_ in kernels), key=lambda k_: by_valu[k_]), "largest share of SQ_INSTS_VALU per proof"
            else:
                name, picked_by = max(kernels.items(), key=lambda kv: kv[1]["total_ms"])[0], "largest shared-chip device time of this run"
            kd = kernels[name]
            per_launch = kd["unit_bytes_per_launch"] or kd["algo_bytes_per_launch"]  # (the unit's bytes where the kernel carries one)
            achieved = per_launch / (kd["avg_launch_ms"] * 1e-3) / 1e9 if kd["avg_launch_ms"] > 0 else 0.0
            roofline = {
                "bound": "hbm", "kernel": name, "kernel_picked_by": picked_by, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": kd["hbm_bytes_per_launch"],
                "avg_launch_ms": kd["avg_launch_ms"], "algo_bytes_per_launch": per_launch,
                "share_of_device_time": kd["total_ms"] / device_ms if device_ms else 0.0,
                "launches_per_proof": kd["launches"] / (args.steps * proofs_per_step),
                "proofs_per_launch": batch,
                "durations": "timed region: twelve provers share the chip (shared-chip durations are not additive); `serialised` = alone on the chip",
                "families": families,
                "note": "BASELINE asks for the HBM roofline; the kernels are integer-ALU bound "
                        "(254-bit Montgomery products), see DESIGN.md and `valu`",
            }
            if serial:
                sk = serial["kernels"][name]
                ach1 = per_launch / (sk["avg_launch_ms"] * 1e-3) / 1e9 if sk["avg_launch_ms"] > 0 else 0.0
                serial.update({"kernel": name, "achieved": ach1, "frac": ach1 / HBM_PEAK_GBPS, "avg_launch_ms": sk["avg_launch_ms"]})
                roofline["serialised"] = serial
        valu = None
        if pmc and "valu" in pmc:
            per_proof = float(pmc["valu"]["wave_instructions_per_proof"])
            ach = per_proof / (ms_per_proof * 1e-3)
            # three yardsticks: the architectural issue peak (one wave64 instruction per SIMD every 2 cycles: only
            # v_mov-class instructions reach it), the 4-cycle rate the SQ counters price a VALU instruction at
            # (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU quad-cycles) at the clock the counter passes measured under this load,
            # and the rate a pure stream of nine-limb Montgomery products sustains (tools/fp64_probe.hip: 173.9 G products/s
            # x 236 instructions)
            sq = load_json("sq_issue.json") or {}
            alone = sq.get("kernels", {})
            # (the timed region's clock: the VALU-bound kernels' -- the time-weighted mean of the counter passes' per-kernel
            #  clocks reads high because the short latency-bound launches do)
            heavy = [v for k_, v in alone.items() if v.get("valu_issue_util", 0) >= 0.7 and v.get("clock_GHz")]
            clock = (sum(v["clock_GHz"] * v["us_per_proof"] for v in heavy) / sum(v["us_per_proof"] for v in heavy)) if heavy else 2.1
            four_cycle = 256 * 4 * clock * 1e9 / 4
            product_loop = 173.9e9 * 236 / 64
            valu = {"wave_instructions_per_proof": per_proof, "achieved_wave_instr_per_s": ach,
                    "issue_peak_wave_instr_per_s": VALU_ISSUE_PEAK, "frac": ach / VALU_ISSUE_PEAK,
                    "peak_note": "one wave64 VALU instruction per SIMD every 2 cycles (SIMD-32), 256 CUs x 4 SIMDs x 2.4 GHz",
                    "clock_GHz_under_load": round(clock, 3),
                    "clock_source": f"{sq.get('_from')}: GRBM_GUI_ACTIVE / duration of the kernels at >= 0.7 issue utilisation, time-weighted",
                    "four_cycle_issue_rate_wave_instr_per_s": four_cycle, "frac_of_four_cycle_issue_rate": ach / four_cycle,
                    "nine_limb_product_loop_rate_wave_instr_per_s": product_loop,
                    "frac_of_nine_limb_product_loop_rate": ach / product_loop,
                    "per_kernel_alone": {k_: round(v["valu_issue_util"], 3) for k_, v in alone.items() if "valu_issue_util" in v},
                    "per_kernel_alone_note": "VALU issue utilisation (4-cycle rate) of each kernel ALONE on the chip, from the counter passes",
                    "source": pmc["valu"].get("source"), "counter_files": [pmc.get("_from"), sq.get("_from")]}
        cs = circuit.cs
        out = {
            "metric": f"create_proof proofs/hour, {circuit.model_name}",
            "value": proofs_per_hour, "unit": "proofs/hour", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "u32x8 (254-bit Montgomery integers)",
            "data": ("checked-in model + benches/example_image_7.png (fixtures from the reference checkout), "
                     "seeded SRS" if args.model != "large" else "synthetic (seeded stand-in model), seeded SRS"),
            "config": {"workload": f"full create_proof of zero_g's WnnCircuit for {circuit.model_name} on "
                                   f"example_image_7.png ({cs.n_advice} advice, {cs.n_fixed} fixed, {len(cs.gates)} gates, "
                                   f"{len(cs.lookups)} lookups, {len(cs.perm_columns)} equality columns, degree {cs.degree()}), "
                                   f"k={circuit.k}, EvaluationDomain's extended domain 2^{cs.extended_k()} (the throughput form "
                                   f"takes the same quotient from {cs.degree() - 1}n points on two cosets), proof {len(streams[0].last[0])} B"
                                   + (" [seeded stand-in model: the file is absent from the reference]"
                                      if args.model == "large" else ""),
                       "class_scores": circuit.scores,
                       "proofs_per_step": proofs_per_step,
                       "parallelism": (f"{world} rank(s), commitments of every proof sharded by point range (one all-gather per "
                                       f"commitment phase and prover, "
                                       f"{'inside the library on RCCL' if exchange_kind == 'rccl' else 'host callback on ' + backend}), "
                                       f"{nprov} prover(s) per rank x lock-step batches of {batch} proofs" if sharded else
                                       f"{world} GPU(s) x {nprov} prover stream(s) x lock-step batches of {batch} proofs")},
            "mode": args.mode if world > 1 else "single-gpu",
            "exchange": exchange_kind, "rccl_ranks": rccl_ranks,
            "collective": {"backend": ("rccl" if backend == "nccl" else backend) if dist is not None else None,
                           "ranks_seen": collective_ranks if dist is not None else None,
                           "how": "all-reduce of ones over the default process group before the timed region"},
            "ranks_share_a_device": bool(world > 1 and "ZG_BENCH_DEVICE" in os.environ),
            "inputs": "host memory, uploaded per proof" if HOST_ADVICE else "resident in HBM",
            "proofs_per_step": proofs_per_step, "ms_per_proof": ms_per_proof,
            "create_proof_wall_s": latency_s, "lone_proof_digit_table_bytes": table_bytes,
            "create_proof_wall_ms_samples": lone_samples_ms, "lone_proof_gate": lone_gate, "runtime_env": RUNTIME_ENV, "provers_per_gpu": nprov, "batch": batch,
            "launches_per_proof": launches_per_proof,
            # SURVEY.md 8d's per-proof figure, twice: from its formula and as the library charged it launch by launch
            # (the units of the msm / ntt / evaluate_h / products families, each once) -- the two must agree
            "algorithmic_bytes_per_proof": algorithmic_bytes_per_proof(cs),
            "algorithmic_bytes_per_proof_charged": charged_unit_bytes / max(1, args.steps * proofs_per_step) if stats else None,
            "algorithmic_GBps": algorithmic_bytes_per_proof(cs) / (ms_per_proof * 1e-3) / 1e9,
            "device_ms_per_proof": device_ms / max(1, args.steps * proofs_per_step),
            "roofline": roofline, "valu": valu,
            "kernels": {k_: {"avg_launch_ms": round(v["avg_launch_ms"], 4), "share": round(v["total_ms"] / device_ms, 4) if device_ms else 0,
                             "algo_GBps": round(v["algo_GBps"], 1), "frac_of_hbm_peak": round(v["algo_GBps"] / HBM_PEAK_GBPS, 5),
                             "hbm_bytes_per_launch": v["hbm_bytes_per_launch"]}
                        for k_, v in sorted(kernels.items(), key=lambda kv: -kv[1]["total_ms"])},
            "single_proof_phase_ms": dict(zip(["advice", "lookups_permuted", "products", "h", "evals", "gwc",
                                               "total", "host_sort"], [round(x, 3) for x in phases])),
        }
        if not args.no_verify:
            out.update(verify_last_step(circuit, streams, host_cores()))  # (a sharded rank holds whole proofs: same check)
    if rank == 0 and not args.no_image_to_proof and not sharded and args.model == "tiny" and world == 1 and not HOST_ADVICE:
        out["image_to_proof"] = image_to_proof(circuit, streams, ctxs, barrier, host_cores(), not args.no_verify)
    # the other three models of BASELINE.json: a few steps each, same driver (after the headline's timed region)
    if not args.no_other_configs and not sharded and args.model == "tiny" and world == 1:
        others = {}
        for s in streams:
            s.prover.close()
        for x in ctxs[1:]:
            x.close()
        ctxs = ctxs[:1]
        # (the headline model's base sets go too: the lone-proof probe left its digit tables on them, 3 x 26 GB at k = 14,
        #  and the k = 17 run below wants 12 provers x 8 slots x 1.4 GiB)
        circuit.g_bases.free()
        circuit.gl_bases.free()
        for m in ("small", "medium", "large"):
            c2 = Circuit(ctx0, m)
            # (k = 15 slots are 0.34 GiB, k = 17 slots 1.4 GiB: batches of 16 and 8 keep 12 provers inside the 288 GB)
            b2 = min(batch, 16) if m != "large" else min(batch, 8)
            np2 = min(nprov, 12)  # (a k = 17 slot is 1.4 GiB: 12 provers x 8 slots + workspaces = 175 GB of the 288)
            cx, st2, (lat, _, tb2) = make_streams(dev_index, c2, ctx0, np2, b2, rank, probe=latency_probe)
            dt2, _ = measure(st2, cx, 3, 1, barrier)
            others[m] = {"model": c2.model_name, "k": c2.k, "ms_per_proof": dt2 / (3 * np2 * b2) * 1e3,
                         "create_proof_wall_s": lat, "lone_proof_digit_table_bytes": tb2, "batch": b2, "provers": np2,
                         "proofs_per_hour": 3 * np2 * b2 / dt2 * 3600.0}
            if m != "large" and not args.no_image_to_proof:  # (the stand-in's program: 360 000 operations, 5 s to record)
                i2p = image_to_proof(c2, st2, cx, barrier, host_cores(), not args.no_verify, steps=2, checks=1)
                others[m]["image_to_proof"] = {k_: i2p[k_] for k_ in ("ms_per_proof", "witness_program", "verified") if k_ in i2p}
            for s in st2:
                s.prover.close()
            c2.g_bases.free()
            c2.gl_bases.free()
            for x in cx[1:]:
                x.close()
        out["other_configs"] = others
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(circuit, len(streams[0].last[0]), host_cores())
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()  # (rank 0 checks its proofs against the oracle after the timed region: the others wait here)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
