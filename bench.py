#!/usr/bin/env python3
"""bench.py -- Wnn::proof (image -> proof) of zero_g's WNN circuit on MI355X (BASELINE.json metric).

The timed region is the reference's own (benches/bench.rs:30-36 times `wnn.proof(&pk, &kzg_params, &img)`, i.e.
/root/reference/src/wnn.rs:232-262: the witness of ONE image and its create_proof): a "step" is `--provers` (default 12)
lock-step batches of `--batch` (default 32) calls of zg_prover_prove_images -- image bytes in, the recorded witness program
replayed on the device (csrc/witness.hip), then the full create_proof (30 commitment MSMs, 21 iNTT + 21 coset NTT + 1 extended
iNTT, 4 lookup arguments, the 2-set permutation argument, evaluate_h, 67 evaluations, 4 GWC openings, Keccak-256
EvmTranscript), proof bytes and class scores out.  Every proof of a step is for ANOTHER image.  The SRS tables and the
proving key are in HBM before the timed region, as in the reference's bench.  K steps = K x provers x batch proofs;
value = proofs / hour over all ranks.  What is timed is what is checked: proofs of the LAST timed step are compared byte
for byte with the oracle's create_proof of the host-synthesised witness of the same image, and pairing-verified.

    python bench.py --gpus N --steps K --warmup W [--mode replicas|shard-msm|msm-only] [--model tiny|small|medium|large]

stdout carries ONE compact JSON line (< 4 KB: the contract keys, `roofline`, `cpu_baseline`, the lone-proof latencies, one
figure per other model); everything else -- per-kernel tables, families, samples, prose -- goes to bench_detail.json next to
this file (and to gpurun_out/bench_detail.json when that directory exists).  The headline is written to stderr and to the
detail file as soon as the timed region ends; every later leg (checks, lone-proof probes, the other three models, the CPU
baseline) runs in its own try block and can only add keys -- a failing leg is named in `errors`, the line still comes out.

  --mode replicas (default): N > 1 runs independent proofs per GPU (weak scaling, no data-path collective).
  --mode shard-msm: the commitments of every proof are sharded by point range over the N GPUs (zg_prover_set_shard: one
      all-gather of partial sums per commitment phase); transforms and evaluate_h stay per GPU.
  --mode msm-only: the commitment MSMs by themselves (north_star: "near-linear MSM scaling 1->8 GPUs"): per rank the
      scalar slices of batch x 30 vectors of the model's shape against its point range of both base sets, ONE all-gather of
      the partial sums, local EC additions; value = MSMs / s over all ranks.
"""
import argparse
import atexit
import json
import os
import signal
import sys
import threading
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
sys.path.insert(0, os.path.join(ROOT, "harness"))


def _arg_gpus(argv) -> int:
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def runtime_env_for(argv, env) -> dict:
    """The HIP runtime settings this process makes for itself (read when the runtime initialises, so before `import torch`).
    GPU_MAX_HW_QUEUES=16: ROCm multiplexes a process's streams onto 4 hardware queues by default; every prover stream
    should have one of its own.  HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory: -2.4 % on a lone proof's 73
    launches) and HSA_ENABLE_INTERRUPT=0 (completion signals polled: -1 %) help ONLY the lone-proof probe, were validated on
    one GPU only (profiles/r04/ab_runtime_env.txt) and have never run beside RCCL's proxy threads: they are made for a
    single-GPU process alone -- never for a rank of an N > 1 run, never for the launcher (whose environment the ranks
    inherit) -- and ZG_BENCH_PLAIN_ENV=1 leaves the runtime exactly as it comes (the `lone.default` probe's child)."""
    if env.get("ZG_BENCH_PLAIN_ENV") == "1":
        return {}
    out = {"GPU_MAX_HW_QUEUES": "16"}
    if int(env.get("WORLD_SIZE", "1")) == 1 and _arg_gpus(argv) == 1:
        out.update({"HIP_FORCE_DEV_KERNARG": "1", "HSA_ENABLE_INTERRUPT": "0"})
    return out


for _k, _v in runtime_env_for(sys.argv[1:], os.environ).items():
    os.environ.setdefault(_k, _v)
RUNTIME_ENV = {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG", "HSA_ENABLE_INTERRUPT")}

import numpy as np
import torch

import wnn_circuit
import wnn_model
import zg_halo2 as zg

R = zg.FR_MODULUS
MONT = (1 << 256) % R
PROFILES = os.path.join(ROOT, "profiles", os.environ.get("ZG_BENCH_PROFILES", "r05"))  # the counter files the line reads
DETAIL = os.path.join(ROOT, "bench_detail.json")
LINE_LIMIT = 4096  # bytes of the stdout line (round 4's 20.7 KB line could not be parsed by the harness)
# the four configurations of BASELINE.json: (k, model); "large" is a seeded stand-in of the same shape
# because model_49input_8192entry_4hash_6bpi.hdf5 is not in the reference checkout (.MISSING_LARGE_BLOBS)
MODELS = {"tiny": wnn_model.MNIST_TINY, "small": wnn_model.MNIST_SMALL, "medium": wnn_model.MNIST_MEDIUM,
          "large": wnn_model.MNIST_LARGE}
DEFAULT_BATCH = {"tiny": 32, "small": 16, "medium": 16, "large": 8}  # (a k = 15 slot is 0.34 GiB, a k = 17 slot 1.4 GiB)
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
        self._program = None

    def witness_program(self):
        """(arrays for zg_witness_plan_create, summary): WnnChip::predict recorded once on a symbolic image
        (harness/witness_tape.py) -- the program zg_prover_prove_images replays per image."""
        if self._program is None:
            import witness_tape

            t0 = time.perf_counter()
            prog = witness_tape.trace(self.wnn, self.k)
            arrays = prog.arrays()
            self._program = (arrays, {"operations": int(arrays["ops"].shape[0]), "levels": int(arrays["level_start"].shape[0] - 1),
                                      "assigned_cells": len(prog.cells), "recorded_in_s": round(time.perf_counter() - t0, 2)})
        return self._program

    def free_bases(self):
        self.g_bases.free()
        self.gl_bases.free()


HOST_ADVICE = False  # --host-advice: measure the PCIe-inclusive rate (never the headline `value`)


class Stream:
    """One proof stream: a prover (own context = own HIP stream + workspace; proving key shared with the others) with
    `batch` slots; step() = one lock-step batch: zg_prover_prove_images (image -> proof) once enable_images() was called,
    else create_proof from the advice columns the slots already hold."""

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
        # columns and reads the rest, so the slots can be proved from again
        self.seed0 = 1_000_000 * stream_id
        self.stream_id = stream_id
        self.steps = 0
        self.plan = None
        self.last_images = None
        self.last_inst = None
        self.last = self.prover.prove_batch([c.advice] * batch, [c.instance] * batch, self.seeds())[0]

    def enable_images(self, arrays: dict, pool: np.ndarray):
        """From here on every proof of a step is for ANOTHER image of `pool`: Wnn::proof through ONE entry point,
        zg_prover_prove_images -- image bytes in, proof bytes and class scores out, inside step()."""
        self.plan = zg.WitnessPlan(self.ctx, arrays)
        self.pool = pool

    def disable_images(self):
        if self.plan is not None:
            self.plan.close()
            self.plan = None
            self.last_images = None
            # (the slots now hold other images' witnesses: put the bench image's back for the from-resident legs)
            self.prover.prove_batch([self.c.advice] * self.batch, [self.c.instance] * self.batch, self.seeds())

    def seeds(self):
        return [self.seed0 + 1000 * self.steps + b for b in range(self.batch)]

    def step(self):
        self.steps += 1
        self.last_seeds = self.seeds()
        if self.plan is not None:
            first = (self.stream_id * 7 + self.steps * self.batch) % len(self.pool)
            self.last_images = [(first + b) % len(self.pool) for b in range(self.batch)]
            self.last, self.last_inst, _ = self.prover.prove_images(self.plan, self.pool[self.last_images], self.last_seeds)
        elif HOST_ADVICE:  # the host-pointer entry: every proof's columns cross PCIe (3 MiB per proof at k = 14)
            self.last = self.prover.prove_batch([self.c.advice] * self.batch, [self.c.instance] * self.batch, self.last_seeds)[0]
        else:
            self.last = self.prover.prove_batch(None, [self.c.instance] * self.batch, self.last_seeds, device=True)[0]
        return self.last


def make_streams(dev_index: int, c: Circuit, ctx0: zg.Ctx, nprovers: int, batch: int, rank: int, exchanges=None, shard=(0, 1),
                 lone: bool = False):
    """The provers of one GPU: the first, then its forks (same proving key and base tables) on contexts of their own.
    ORDER MATTERS: HIP hands out hardware queues in stream-creation order and the chip runs four compute pipes, so
    streams whose queue indices are equal mod 4 share a pipe.  The throughput streams are therefore created first and
    back to back, and the lone-proof prover (`lone`: a further fork, batch of one) after them, so that its main and side
    stream land on two different pipes again (3.0 ms; 3.8 ms when the pair shares one).
    Returns (contexts of the throughput streams, the streams, the lone Stream or None)."""
    base = 0 if shard[1] > 1 else rank * 64  # (ranks of a sharded proof prove the SAME statements with the same keys)
    ctxs = [ctx0] + [zg.Ctx(dev_index) for _ in range(nprovers - 1)]
    first = zg.Prover(ctx0, c.img, c.fixed, c.sigma, c.g_bases, c.gl_bases, c.vk_repr)
    first.set_overlap(False)  # (before forking: a fork of a single-stream prover creates no side stream of its own)
    provers = [first] + [first.fork(x) for x in ctxs[1:]]
    streams = [Stream(ctxs[i], provers[i], c, batch, base + i, exchanges[i] if exchanges else None, shard) for i in range(nprovers)]
    lone_stream = None
    if lone:
        pctx = zg.Ctx(dev_index)
        lone_stream = Stream(pctx, first.fork(pctx), c, 1, base + 63)
    return ctxs, streams, lone_stream


def close_streams(ctxs, streams, lone_stream=None):
    for s in streams + ([lone_stream] if lone_stream else []):
        if s.plan is not None:
            s.plan.close()
            s.plan = None
        s.prover.close()
    if lone_stream:
        lone_stream.ctx.close()
    for x in ctxs[1:]:
        x.close()


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


class PowerSampler:
    """Package power and shader clock of this box's GPU(s) while the timed steps run, read from the driver's hwmon files
    (/sys/class/drm/card*/device/hwmon/hwmon*/{power1_input, freq1_input, power1_cap}: no process, no HIP call) every 50 ms
    by one thread: says whether the figure was made at the part's power limit and at which clock.  Best effort: an empty
    object where the files are not readable."""

    def __init__(self, period: float = 0.05):
        import glob
        cards = [os.path.dirname(f) for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))]
        # only THIS process's device (a shared host shows its neighbours' cards as well): matched by PCI address
        mine = self._pci_of_current_device()
        own = [c for c in cards if mine and os.path.basename(os.path.realpath(os.path.join(c, "..", ".."))).lower() == mine]
        self.cards, self.matched = (own, True) if own else (cards, False)
        self.period, self.rows, self.stop_flag, self.thread = period, [], threading.Event(), None

    @staticmethod
    def _pci_of_current_device():
        try:
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            return "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:  # noqa: BLE001  (no such attributes on this build: every busy card is reported, flagged unmatched)
            return None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def start(self):
        if not self.cards:
            return self

        def work():
            while not self.stop_flag.is_set():
                self.rows.append([(self._read(c + "/power1_input"), self._read(c + "/freq1_input")) for c in self.cards])
                self.stop_flag.wait(self.period)

        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()
        return self

    def stop(self) -> dict:
        if not self.thread:
            return {}
        self.stop_flag.set()
        self.thread.join()
        out = []
        for i, c in enumerate(self.cards):
            pw = [r[i][0] / 1e6 for r in self.rows if r[i][0] is not None]
            fq = [r[i][1] / 1e6 for r in self.rows if r[i][1] is not None]
            if not pw or max(pw) < 400.0:  # (an idle card of the node: 245 W)
                continue
            cap = self._read(c + "/power1_cap")
            out.append({"card": next((x for x in c.split("/") if x.startswith("card")), "?"), "samples": len(pw), "power_w_avg": round(sum(pw) / len(pw), 1), "power_w_max": round(max(pw), 1),
                        "power_cap_w": round(cap / 1e6, 1) if cap else None,
                        "sclk_mhz_avg": round(sum(fq) / len(fq), 1) if fq else None, "sclk_mhz_min": round(min(fq), 1) if fq else None,
                        "sclk_mhz_max": round(max(fq), 1) if fq else None})
        if not self.matched and len(out) > 1:  # (could not tell which card is ours: keep the busiest one only)
            out = [max(out, key=lambda x: x["power_w_avg"])]
        return {"source": "hwmon power1_input / freq1_input every %d ms over the timed steps" % int(self.period * 1e3),
                "matched_by_pci_address": self.matched, "cards": out}


LAST_POWER = {}


def measure(streams, ctxs, steps, warmup, barrier, profile=False):
    """`warmup` untimed steps, then EXACTLY `steps` timed ones between two barriers; with `profile` every launch carries
    its own start / stop event on the stream it is launched on (hipExtLaunchKernelGGL), collected per kernel afterwards."""
    if WATCHDOG:
        WATCHDOG.arm(True)
    run_steps(streams, max(warmup, 1))
    for x in ctxs:
        x.profile(profile)
    barrier()
    sampler = PowerSampler().start()
    t0 = time.perf_counter()
    run_steps(streams, steps)
    barrier()
    dt = time.perf_counter() - t0
    LAST_POWER.clear()
    LAST_POWER.update(sampler.stop())
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


def algorithmic_bytes_per_proof(cs) -> float:
    """MSM n*(32+64)+96, NTT 2*n*32, coeff->ext (n+8n)*32, ext->coeff 2*8n*32,
    grand product 3*n*32, evaluate_h (inputs+1)*8n*32."""
    n, en = 1 << cs.k, 1 << cs.extended_k()
    sets = (len(cs.perm_columns) + cs.degree() - 3) // (cs.degree() - 2)
    nl = len(cs.lookups)
    msm = commitments_per_proof(cs) * (n * 96 + 96)
    polys = cs.n_advice + cs.n_instance + 3 * nl + sets
    intt = polys * 2 * n * 32
    ext = polys * (n + en) * 32
    ext_inv = 2 * en * 32
    gp = (nl + sets) * 3 * n * 32
    eh_inputs = cs.n_advice + cs.n_instance + cs.n_fixed + len(cs.perm_columns) + sets + 3 * nl + 3
    eh = (eh_inputs + 1) * en * 32
    return float(msm + intt + ext + ext_inv + gp + eh)


def commitments_per_proof(cs) -> int:
    """advice + 3 per lookup + permutation sets + the random polynomial + the h pieces + 4 GWC quotients (30 for the WNN)"""
    sets = (len(cs.perm_columns) + cs.degree() - 3) // (cs.degree() - 2)
    return cs.n_advice + 3 * len(cs.lookups) + sets + 1 + (cs.degree() - 1) + 4


FAMILIES = {"msm": ("msm_",), "ntt": ("ntt_",), "evaluate_h": ("evaluate_h",), "sort": ("sort_", "permute_"),
            "products": ("grand_product", "lookup_", "perm_terms", "permuted_finish", "blind_rows", "random_poly"),
            "openings": ("eval_dot", "powers", "horner_combine", "kate_", "fold", "diff_scale", "split_combine"),
            "witness": ("witness_",)}  # (ZG_LAUNCH labels)


def family_of(kernel: str) -> str:
    for fam, prefixes in FAMILIES.items():
        if kernel.startswith(prefixes):
            return fam
    return "other"


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


def load_pmc(model: str = "tiny"):
    """Counter figures of the same configuration (rocprofv3 --pmc passes, tools/profile.sh + tools/install_profile.py;
    committed under profiles/rNN): HBM bytes per launch per kernel and VALU wave-instructions per proof.  The tiny model's
    files carry no prefix (as in rounds 1-4), the others `<model>_`."""
    return load_json("pmc_traffic.json" if model == "tiny" else f"{model}_pmc_traffic.json")


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


def kernel_table(kernels: dict, device_ms: float) -> dict:
    return {k_: {"avg_launch_ms": round(v["avg_launch_ms"], 4), "share_of_device_time": round(v["total_ms"] / device_ms, 4) if device_ms else 0.0,
                 "algo_GBps": round(v["algo_GBps"], 1), "frac_of_hbm_peak": round(v["algo_GBps"] / HBM_PEAK_GBPS, 5),
                 "hbm_bytes_per_launch": v["hbm_bytes_per_launch"]}
            for k_, v in sorted(kernels.items(), key=lambda kv: -kv[1]["total_ms"])}


def serialised_pass(streams, ctxs, barrier, batch: int, pmc, pmc_scale: float, steps: int = 3) -> dict:
    """ONE prover stepping, same lock-step batches, every launch bracketed by its own events -- one stream, so the kernels run
    one at a time: what a kernel costs ALONE on the chip.  The timed region's durations are shared-chip durations (eleven
    other provers run beside every launch) and flip from box to box; the dominant kernel is picked HERE."""
    dt1, st1 = measure(streams[:1], ctxs[:1], steps, 1, barrier, profile=True)
    k1, f1, dev1, _ = roofline_tables(st1, pmc, pmc_scale)
    return {"note": "one prover alone on the chip (one stream: kernels run one at a time), same batches of "
                    f"{batch}; per-launch HIP events; algorithmic bytes / launch duration against the HBM peak",
            "ms_per_proof": dt1 / (steps * batch) * 1e3, "device_ms_per_proof": dev1 / (steps * batch),
            "kernels": kernel_table(k1, dev1), "families": f1, "_raw": k1}


def roofline_object(stats, serial, pmc, pmc_scale, n_proofs: int, batch: int):
    """The `roofline` object of the line: the dominant kernel (largest device time of the SERIALISED pass; without one the
    largest share of the proof's VALU instructions from the counter file, else of this run's shared-chip time), its
    algorithmic bytes per launch (the SURVEY 8d unit where it carries one) over its average launch duration in the TIMED
    region (`frac`: twelve provers share the chip) and alone on the chip (`serialised.frac`), against the HBM peak."""
    kernels, families, device_ms, charged = roofline_tables(stats, pmc, pmc_scale)
    by_valu = (pmc or {}).get("valu", {}).get("by_kernel") or {}
    if serial:
        name, picked_by = next(iter(serial["kernels"])), "largest device time in the serialised pass (one prover alone on the chip)"
    elif [k_ for k_ in by_valu if k_ in kernels]:
        name, picked_by = max((k_ for k_ in by_valu if k_ in kernels), key=lambda k_: by_valu[k_]), "largest share of SQ_INSTS_VALU per proof"
    else:
        name, picked_by = max(kernels.items(), key=lambda kv: kv[1]["total_ms"])[0], "largest shared-chip device time of this run"
    kd = kernels.get(name) or (serial and serial["_raw"][name])
    per_launch = kd["unit_bytes_per_launch"] or kd["algo_bytes_per_launch"]  # (the unit's bytes where the kernel carries one)
    achieved = per_launch / (kd["avg_launch_ms"] * 1e-3) / 1e9 if kd["avg_launch_ms"] > 0 else 0.0
    msm = families.get("msm", {})
    roofline = {
        "bound": "hbm", "kernel": name, "kernel_picked_by": picked_by, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS, "traffic": kd["hbm_bytes_per_launch"],
        "avg_launch_ms": kd["avg_launch_ms"], "algo_bytes_per_launch": per_launch,
        "share_of_device_time": kd["total_ms"] / device_ms if device_ms else 0.0,
        "launches_per_proof": kd["launches"] / max(1, n_proofs), "proofs_per_launch": batch,
        "msm_counter_over_algorithmic": msm.get("counter_over_algorithmic_bytes"),
        "counter_file": (pmc or {}).get("_from"),
        "durations": "timed region: the provers share the chip (shared-chip durations are not additive); `serialised` = alone on the chip",
        "families": families,
        "note": "BASELINE asks for the HBM roofline; the kernels are integer-ALU bound (254-bit Montgomery products), see DESIGN.md and `valu`",
    }
    if serial:
        sk = serial["_raw"][name]
        ach1 = per_launch / (sk["avg_launch_ms"] * 1e-3) / 1e9 if sk["avg_launch_ms"] > 0 else 0.0
        ser = {k_: v for k_, v in serial.items() if k_ != "_raw"}
        ser.update({"kernel": name, "achieved": ach1, "frac": ach1 / HBM_PEAK_GBPS, "avg_launch_ms": sk["avg_launch_ms"]})
        roofline["serialised"] = ser
    return roofline, kernels, device_ms, charged


def valu_object(pmc, ms_per_proof: float):
    """three yardsticks: the architectural issue peak (one wave64 instruction per SIMD every 2 cycles: only v_mov-class
    instructions reach it), the 4-cycle rate the SQ counters price a VALU instruction at (SQ_ACTIVE_INST_VALU ==
    SQ_INSTS_VALU quad-cycles) at the clock the counter passes measured under this load, and the rate a pure stream of
    nine-limb Montgomery products sustains (tools/fp64_probe.hip: 173.9 G products/s x 236 instructions)"""
    if not pmc or "valu" not in pmc:
        return None
    per_proof = float(pmc["valu"]["wave_instructions_per_proof"])
    ach = per_proof / (ms_per_proof * 1e-3)
    sq = load_json(os.path.basename(pmc["_from"]).replace("pmc_traffic", "sq_issue")) or {}
    alone = sq.get("kernels", {})
    # (the timed region's clock: the VALU-bound kernels' -- the time-weighted mean of the counter passes' per-kernel
    #  clocks reads high because the short latency-bound launches do)
    heavy = [v for v in alone.values() if v.get("valu_issue_util", 0) >= 0.7 and v.get("clock_GHz")]
    clock = (sum(v["clock_GHz"] * v["us_per_proof"] for v in heavy) / sum(v["us_per_proof"] for v in heavy)) if heavy else 2.1
    four_cycle = 256 * 4 * clock * 1e9 / 4
    product_loop = 173.9e9 * 236 / 64
    return {"wave_instructions_per_proof": per_proof, "achieved_wave_instr_per_s": ach,
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


# ------------------------------------------------------------------------------------------------ checks (the oracle)

def _oracle_pk(c: Circuit, threads: int):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc

    orc.load().orc_set_threads(threads)
    params = orc.params_from_scalar(c.k, c.s)
    return orc, orc.ProvingKey(c.img, c.fixed, c.sigma, params, c.vk_repr)


def verify_last_step(c: Circuit, streams, threads: int, byte_checks: int = 3) -> dict:
    """What was timed is what is checked: the proofs the streams made in the LAST timed step against the oracle.
    Image -> proof steps: byte for byte against the oracle's create_proof of the HOST-synthesised witness of the same image
    (harness/wnn_circuit.py) for the first and the last proof of stream 0's batch and the first of the last stream; the
    class scores of those against the model mirror's predict; the public pairing equation for the first proof of every
    other stream.  From-resident steps: byte for byte against the oracle's proof of the bench witness."""
    orc, pk = _oracle_pk(c, threads)
    picks = list(dict.fromkeys([(0, 0), (0, streams[0].batch - 1), (len(streams) - 1, 0)]))[:byte_checks]
    shape = wnn_model.load_test_image().shape
    checked = 0
    for i, b in picks:
        s = streams[i]
        if s.last_images is not None:
            im = s.pool[s.last_images[b]].reshape(shape)
            _, asg, ilen, scores = wnn_circuit.build(c.wnn, im, c.k)
            adv, inst = asg.advice_values(), asg.instance_values(ilen)
            if not np.array_equal(inst[0], s.last_inst[b]) or scores != c.wnn.predict(im):
                return {"verified": False, "detail": f"stream {i} proof {b}: class scores differ from the model's"}
        else:
            adv, inst = c.advice, c.instance
        st, want, _ = orc.create_proof(pk, adv, inst, s.last_seeds[b])
        if st != 0 or want != s.last[b]:
            return {"verified": False, "detail": f"stream {i} proof {b} differs from the oracle"}
        checked += 1
    paired = 0
    for i in range(len(streams)):
        if (i, 0) in picks and i != len(streams) - 1:
            continue
        s = streams[i]
        inst = s.last_inst[0][None, :, :] if s.last_images is not None else c.instance
        if orc.verify_proof_pairing(pk, inst, s.last[0]) != 1:
            return {"verified": False, "detail": f"pairing check failed for stream {i}"}
        paired += 1
    what = "the host-synthesised witness of the same image" if streams[0].last_images is not None else "the bench witness"
    return {"verified": True, "detail": f"{checked} proofs of the last timed step byte-identical to the oracle's create_proof of {what}, "
                                        f"{paired} more (one per other stream) pairing-verified"}


def cpu_baseline(c: Circuit, proof_len: int, threads: int, repeats: int = 7):
    """The oracle's create_proof (CPU restatement of halo2's algorithms, OpenMP over MSM chunks, FFT butterflies and row
    loops) timed on this box's host cores on the SAME circuit and witness (its own seeded SRS of the same size): the
    MEDIAN of `repeats` proofs after one untimed warm-up, every sample and the per-phase split of
    the median proof in the record.  kind = "port": halo2's own Rust prover cannot be built here (no cargo/rustc).  The
    witness synthesis (which the reference's timed region includes) is NOT in it: the harness synthesises in Python."""
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
        "sample": f"median of {repeats} oracle create_proofs, k={c.k}, OpenMP x{threads}, no witness synthesis; expected SLOWER than real halo2",
        "sample_long": f"median of {repeats} full create_proofs of the same k={c.k} circuit (after 1 warm-up): {dt:.3f} s each; "
                       f"oracle/prover.c (plain-C restatement of halo2 create_proof, OpenMP {threads} threads); its evaluate_h "
                       f"walks the expanded gate monomials row by row as prefix tries (135 products per row at k=14 for the gates) where halo2's "
                       f"GraphEvaluator runs the shared expression graph (the `h` phase is {phases[mid][3] / max(phases[mid][6], 1e-9):.0%} of this proof; the key's "
                       f"polys and cosets are kept with the key as keygen_pk does), on {threads} of the machine's {os.cpu_count()} cores: expected SLOWER than real halo2 on the "
                       f"same cores -- a baseline to be read with that, never a speed-up claim",
        "wall_s": dt, "samples_s": [round(x, 4) for x in samples], "min_s": min(samples), "max_s": max(samples),
        "openmp": {"threads": threads, "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"), "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"),
                   "cores_in_affinity_mask": affinity, "machine_cores": os.cpu_count()},
        "phase_ms": dict(zip(["advice", "lookups_permuted", "products", "h", "evals", "gwc", "total"],
                             [round(x, 2) for x in phases[mid][:7]])),
    }


def image_pool(c: Circuit, count: int = 64) -> np.ndarray:
    """benches/example_image_7.png and seeded noise images of its shape (synthetic: the MNIST test set is not in the
    reference checkout)"""
    real = wnn_model.load_test_image()
    rng = np.random.default_rng(2024)
    return np.stack([real] + [rng.integers(0, 256, size=real.shape, dtype=real.dtype) for _ in range(count - 1)]).reshape(count, -1)


# ------------------------------------------------------------------------------------------------ a lone proof

def under_counter_collection() -> bool:
    """rocprofv3 --pmc serialises kernels across queues: a spinning gate kernel would hold back the side-stream work the
    host waits for (every gated proof would stall for the gate's time limit and be made twice) -- the probe then keeps the
    gate off and says so."""
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def median(xs):
    return sorted(xs)[len(xs) // 2]


def lone_probe(s: Stream, tables: bool, gate: bool, images: bool = True, repeats: int = 9) -> dict:
    """One proof alone on the chip, the latency form (transforms on a side stream, several lanes per EC addition):
    `create_proof` from the advice columns resident in slot 0, and `image_to_proof` = zg_prover_prove_images(count = 1), the
    reference's own timed region for ONE image; each the median of `repeats` single measurements after three warm-ups.
    tables: zg_prover_enable_digit_tables first (78 GB at k = 14, an explicit call); gate: ZG_LAT_GATE=1 (opt-in: each
    phase queued behind a kernel that waits for the host's challenge).  The gated proofs' bytes are compared with the
    same seeds proved again in the plain order, and `witness_run` / `witness_finish` are timed alone."""
    p, c, ctx = s.prover, s.c, s.ctx
    p.set_overlap(True)
    out = {"digit_table_bytes": p.enable_digit_tables() if tables else 0}
    gate = bool(gate and not under_counter_collection())
    out["gate"] = gate
    plan, img = None, None
    if images:
        arrays, _ = c.witness_program()
        plan = zg.WitnessPlan(ctx, arrays)
        img = wnn_model.load_test_image().reshape(1, -1)
    if gate:
        zg.tuning_set("ZG_LAT_GATE", 1)
    try:
        for _ in range(3):
            p.prove_dev(p.advice_slot(0), c.instance, 1)
        each, proof = [], None
        for i in range(repeats):  # (one proof per measurement)
            t0 = time.perf_counter()
            proof = p.prove_dev(p.advice_slot(0), c.instance, 2 + i)
            each.append((time.perf_counter() - t0) * 1e3)
        out["create_proof_ms"] = median(each)
        out["create_proof_ms_samples"] = [round(x, 4) for x in each]
        out["phase_ms"] = dict(zip(["advice", "lookups_permuted", "products", "h", "evals", "gwc", "total", "host_sort"],
                                   [round(x, 3) for x in p.phase_ms()]))
        iproof = inst = None
        if plan is not None:
            for _ in range(3):
                p.prove_images(plan, img, [1])
            each = []
            for i in range(repeats):
                t0 = time.perf_counter()
                proofs, inst, _ = p.prove_images(plan, img, [2 + i])
                each.append((time.perf_counter() - t0) * 1e3)
            iproof = proofs[0]
            out["image_to_proof_ms"] = median(each)
            out["image_to_proof_ms_samples"] = [round(x, 4) for x in each]
    finally:
        if gate:
            zg.tuning_set("ZG_LAT_GATE", -1)
    # the same seeds in the plain order: proofs are deterministic per (witness, key), whatever the schedule
    last = 2 + repeats - 1
    same = p.prove_dev(p.advice_slot(0), c.instance, last) == proof
    if plan is not None:
        again, inst2, _ = p.prove_images(plan, img, [last])
        same = same and again[0] == iproof and iproof == proof and np.array_equal(inst, inst2) and np.array_equal(inst[0], c.instance[0])
        ctx.profile(True)
        for _ in range(3):
            p.prove_images(plan, img, [1])
        st = ctx.profile_collect()
        ctx.profile(False)
        for k_ in ("witness_run", "witness_finish"):
            if k_ in st and st[k_][0]:
                out[k_ + "_ms"] = st[k_][1] / st[k_][0]
        plan.close()
    out["bytes_equal_plain_order"] = bool(same)
    p.set_overlap(False)
    return out


def lone_child(args) -> int:
    """`bench.py --lone-child MODEL`: the lone-proof pair as the library and the runtime COME -- no digit tables, no gate,
    and (the parent starts this process with ZG_BENCH_PLAIN_ENV=1) none of bench.py's HIP runtime settings.  One prover,
    prints one JSON object."""
    ctx = zg.Ctx(int(os.environ.get("ZG_BENCH_DEVICE", "0")))
    c = Circuit(ctx, args.lone_child)
    p = zg.Prover(ctx, c.img, c.fixed, c.sigma, c.g_bases, c.gl_bases, c.vk_repr)
    s = Stream(ctx, p, c, 1, 63)
    out = lone_probe(s, tables=False, gate=False, images=not args.from_resident)
    out["runtime_env"] = RUNTIME_ENV
    p.close()
    c.free_bases()
    ctx.close()
    print(json.dumps(out), flush=True)
    return 0


def lone_default_in_child(model: str, images: bool, timeout: float = 600.0) -> dict:
    """Runs lone_child in a process of its own (a child, never an exec) whose environment has none of this process's HIP
    runtime settings; this process's streams are idle meanwhile."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG", "HSA_ENABLE_INTERRUPT",
                                                            "ZG_LAT_GATE", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ZG_BENCH_PLAIN_ENV"] = "1"
    cmd = [sys.executable, os.path.abspath(__file__), "--lone-child", model] + ([] if images else ["--no-image-to-proof"])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    if r.returncode != 0:
        raise RuntimeError(f"lone child exited {r.returncode}: {r.stderr[-600:]}")
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


# ------------------------------------------------------------------------------------------------ the line

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config")


def _r(x, digits: int = 6):
    """floats to `digits` significant digits, recursively (the line is for a parser, the detail file keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if d and k in d and d[k] is not None} if d else None


def format_line(out: dict, limit: int = LINE_LIMIT) -> str:
    """THE one line of stdout: the contract's keys, compact `roofline` and `cpu_baseline` objects, the lone-proof latencies,
    one figure per other model and the path of the detail file -- never more than `limit` bytes (optional keys are dropped,
    last first, should strings ever grow; `tests/test_bench_contract.py` builds a line through this function)."""
    c = {k: out.get(k) for k in CONTRACT_KEYS}
    cfg = out.get("config") or {}
    c["config"] = {"workload": str(cfg.get("workload", ""))[:420], "proofs_per_step": cfg.get("proofs_per_step"),
                   "parallelism": str(cfg.get("parallelism", ""))[:200]}
    c["data"] = str(c.get("data") or "")[:160]
    optional = []  # (key, value) in the order they are dropped LAST -> first

    def add(key, value):
        if value is not None and value != {}:
            optional.append((key, value))

    add("mode", out.get("mode"))
    add("ms_per_proof", out.get("ms_per_proof"))
    add("verified", out.get("verified"))
    rf = out.get("roofline")
    if rf:
        r = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "algo_bytes_per_launch", "avg_launch_ms",
                       "msm_counter_over_algorithmic"))
        r["traffic"] = rf.get("traffic")  # (null when no counter file covers this configuration: the key stays)
        if rf.get("serialised"):
            r["serialised"] = _pick(rf["serialised"], ("frac", "avg_launch_ms", "ms_per_proof"))
        add("roofline", r)
    cb = out.get("cpu_baseline")
    if cb:
        b = _pick(cb, ("value", "unit", "cores", "kind", "wall_s"))
        b["sample"] = str(cb.get("sample", ""))[:120]
        add("cpu_baseline", b)
    add("collective", _pick(out.get("collective"), ("backend", "ranks_seen")))
    add("from_resident_columns_ms_per_proof", out.get("from_resident_columns_ms_per_proof"))
    add("create_proof_wall_s", out.get("create_proof_wall_s"))
    add("image_to_proof_wall_s", out.get("image_to_proof_wall_s"))
    lone = out.get("lone") or {}
    lc = {}
    for form in ("opted_in", "default"):
        if lone.get(form):
            lc[form] = _pick(lone[form], ("create_proof_ms", "image_to_proof_ms", "witness_run_ms", "digit_table_bytes", "gate",
                                          "bytes_equal_plain_order"))
    add("lone", lc)
    add("valu", _pick(out.get("valu"), ("frac_of_four_cycle_issue_rate",)))
    pw = (out.get("power") or {}).get("cards") or []
    if pw:  # (the busiest card: is the figure made at the part's power limit, and at which clock)
        add("power", {**_pick(max(pw, key=lambda x: x.get("power_w_avg") or 0), ("power_w_avg", "power_w_max", "power_cap_w", "sclk_mhz_avg")),
                      **_pick(out.get("power"), ("joules_per_proof",))})
    oc = out.get("other_configs") or {}
    add("other_configs", {m: _pick(v, ("ms_per_proof", "image_to_proof_wall_s", "verified")) for m, v in oc.items() if isinstance(v, dict)})
    add("algorithmic_bytes_per_proof", out.get("algorithmic_bytes_per_proof"))
    add("runtime_env", {k: v for k, v in (out.get("runtime_env") or {}).items() if v is not None})
    add("errors", sorted(out.get("errors") or {}) or None)
    add("detail", out.get("detail_file"))
    while True:
        line = json.dumps(_r({**c, **dict(optional)}), separators=(",", ":"))
        if len(line.encode()) < limit or not optional:
            break
        # drop from the end of the list, but keep roofline / cpu_baseline / verified for as long as anything else is left
        keep = ("roofline", "cpu_baseline", "verified", "ms_per_proof")
        victims = [i for i, (k, _) in enumerate(optional) if k not in keep] or list(range(len(optional)))
        optional.pop(victims[-1])
    assert len(line.encode()) < limit, "the contract keys alone exceed the line limit"
    return line


class Emitter:
    """Holds the measurement from the moment the timed region ends.  early(): the compact line to STDERR and the full
    record to the detail file -- before any tail leg starts.  final(): the ONE line of stdout (the contract: one JSON line),
    exactly once -- from main(), or from atexit / SIGTERM should a tail leg take the process down."""

    def __init__(self):
        self.out, self.printed = None, False
        atexit.register(self.final)
        try:
            signal.signal(signal.SIGTERM, self._term)
        except ValueError:  # (not the main thread: a test imports the module)
            pass

    def _term(self, *_):
        self.final()
        os._exit(143)

    def write_detail(self):
        if self.out is None:
            return
        paths = [DETAIL] + ([os.path.join(ROOT, "gpurun_out", "bench_detail.json")] if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else [])
        for path in paths:
            try:
                with open(path + ".tmp", "w") as f:
                    json.dump(self.out, f, indent=1, default=str)
                os.replace(path + ".tmp", path)
            except OSError:
                pass

    def early(self, out: dict):
        self.out = out
        print("bench.py headline (tail legs follow; the final line goes to stdout): " + format_line(out), file=sys.stderr, flush=True)
        self.write_detail()

    def leg(self, name: str, fn):
        """one tail leg: its result (a dict of keys for `out`) or {"error": ...} under out["errors"][name] -- never an exception"""
        t0 = time.perf_counter()
        try:
            got = fn()
            if got:
                self.out.update(got)
        except BaseException as e:  # noqa: BLE001  (KeyboardInterrupt included: the line must still come out)
            import traceback

            self.out.setdefault("errors", {})[name] = {"error": f"{type(e).__name__}: {e}"[:500], "traceback": traceback.format_exc()[-1500:]}
            print(f"bench.py: tail leg `{name}` failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            if isinstance(e, (KeyboardInterrupt, SystemExit)):
                self.final()
                raise
        self.out.setdefault("leg_seconds", {})[name] = round(time.perf_counter() - t0, 2)
        self.write_detail()

    def final(self):
        if self.out is not None and not self.printed:
            self.printed = True
            self.write_detail()
            print(format_line(self.out), flush=True)


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


# ------------------------------------------------------------------------------------------------ ranks

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


def setup_ranks(args) -> SimpleNamespace:
    """One process per GPU: device, process group (RCCL when every rank owns a GPU; gloo for one-GPU rehearsals and the
    GPU-less --dry-run), and the number of ranks the collective backend ITSELF counted."""
    global WATCHDOG
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dry = bool(args.dry_run)
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # (rehearsal knobs for a one-GPU box: ZG_BENCH_DEVICE=0 puts every rank on the same card, ZG_BENCH_BACKEND=gloo
    #  replaces RCCL, which refuses two ranks on one GPU)
    dev_index = int(os.environ.get("ZG_BENCH_DEVICE", local_rank))
    backend = "gloo" if dry else os.environ.get("ZG_BENCH_BACKEND", "nccl")
    dev = None
    if not dry:
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
            if not dry:
                torch.cuda.synchronize()
            # what the collective backend itself saw: an all-reduce of ones over the default group (RCCL when every rank
            # owns a GPU) -- the N of an N-GPU line is this number, not an argument echoed back
            ones = torch.ones(1, dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(ones)
            collective_ranks = int(ones.item())
        if collective_ranks != world:
            raise SystemExit(f"bench.py: the {backend} group counts {collective_ranks} ranks, WORLD_SIZE says {world}")
        WATCHDOG = Watchdog(float(os.environ.get("ZG_BENCH_STALL_S", "300")))
    # every rank on a GPU of its own (the driver's launch): RCCL; ranks sharing a card (rehearsal): host callback
    own_gpu = backend == "nccl" and "ZG_BENCH_DEVICE" not in os.environ
    return SimpleNamespace(world=world, rank=rank, dev_index=dev_index, dev=dev, backend=backend, dist=dist, dry=dry,
                           collective_ranks=collective_ranks, own_gpu=own_gpu,
                           rccl_ranks=collective_ranks if (dist is not None and backend == "nccl") else None,
                           tensor_device=dev if backend == "nccl" else "cpu")


def max_over_ranks(rk, dt: float) -> float:
    if rk.dist is None:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=rk.tensor_device)
    rk.dist.all_reduce(t, op=rk.dist.ReduceOp.MAX)
    return float(t.item())


def collective_object(rk) -> dict:
    return {"backend": ("rccl" if rk.backend == "nccl" else rk.backend) if rk.dist is not None else None,
            "ranks_seen": rk.collective_ranks if rk.dist is not None else None,
            "how": "all-reduce of ones over the default process group before the timed region"}


# ------------------------------------------------------------------------------------------------ proofs (replicas, shard-msm)

def build_job(args, rk) -> SimpleNamespace:
    sharded = args.mode == "shard-msm" and rk.world > 1
    shard = (rk.rank, rk.world) if sharded else (0, 1)
    exchange_kind = (args.exchange or ("rccl" if rk.own_gpu else "host")) if sharded else None
    # shard-msm over raw RCCL communicators: ONE prover per rank unless --provers says otherwise.  Several provers of a
    # process would issue ncclAllGather on several communicators from several threads, in an order that differs from rank
    # to rank -- the classic NCCL/RCCL deadlock once the collectives' kernels cannot all be resident -- and no
    # world > 1 run has ever validated it; the host exchange (gloo / torch.distributed groups, one per prover) has no
    # device-side spinning and keeps 4.
    nprov = max(1, args.provers if args.provers else (12 if not sharded else 1 if exchange_kind == "rccl" else 4))
    batch = max(1, args.batch if args.batch else DEFAULT_BATCH[args.model])
    if args.model == "large" and not args.batch:
        batch = min(batch, 8)  # (a k = 17 proof slot is 1.4 GiB; 12 provers x 8 slots + workspaces stay well inside 288 GB)
    exchanges = None
    # rccl_ranks: the ranks RCCL itself counted -- the all-reduce above when the default group is RCCL (replicas and the
    # host exchange), ncclCommCount of the prover's communicator in shard-msm's in-library exchange; null under gloo
    rccl_ranks = rk.rccl_ranks
    if sharded:
        import multi_gpu

        dist = rk.dist
        # ONE exchange PER PROVER (prover i of every rank forms a group with prover i of the others): the provers of a
        # rank work through their phases independently, each on its own stream / host thread
        with stdout_to_stderr():
            if exchange_kind == "rccl":
                exchanges = [multi_gpu.RcclComm(rk.rank, rk.world, rk.dev_index, dist) for _ in range(nprov)]
                rccl_ranks = exchanges[0].count()
            else:
                groups = [dist.new_group(backend=None) for _ in range(nprov)]  # (collective calls: same order on every rank)
                exchanges = [multi_gpu.make_exchange(dist, rk.dev if rk.backend == "nccl" else None, g) for g in groups]
                for g in groups:  # (a group's transport comes up at its first collective: now, not inside the timed region)
                    dist.barrier(group=g)
    ctx0 = zg.Ctx(rk.dev_index)
    circuit = Circuit(ctx0, args.model, shard)
    want_lone = rk.world == 1 and not sharded and not args.no_latency_probe
    ctxs, streams, lone = make_streams(rk.dev_index, circuit, ctx0, nprov, batch, rk.rank, exchanges, shard, lone=want_lone)
    images = not (args.from_resident or HOST_ADVICE)
    if images:
        arrays, _ = circuit.witness_program()
        pool = image_pool(circuit)
        for s in streams:
            s.enable_images(arrays, pool)

    def barrier():
        if rk.dist is not None:
            rk.dist.barrier()
        torch.cuda.synchronize(rk.dev)
        for x in ctxs:
            x.sync()

    return SimpleNamespace(args=args, rk=rk, sharded=sharded, shard=shard, exchange_kind=exchange_kind, nprov=nprov, batch=batch,
                           rccl_ranks=rccl_ranks, ctx0=ctx0, circuit=circuit, ctxs=ctxs, streams=streams, lone=lone, images=images,
                           barrier=barrier)


def measure_headline(job) -> dict:
    """The timed region (exactly --steps steps after --warmup untimed ones, barrier + synchronize on both sides, the MAX
    over ranks) and the line's contract keys from it.  Nothing here touches the oracle."""
    args, rk, c = job.args, job.rk, job.circuit
    # every launch carries its own start / stop event (hipExtLaunchKernelGGL on the prover's stream): a lock-step batch is
    # ~100 launches for `batch` proofs, so timing them all costs nothing measurable
    dt, stats = measure(job.streams, job.ctxs, args.steps, args.warmup, job.barrier, profile=not args.no_kernel_events)
    dt = max_over_ranks(rk, dt)
    job.stats = stats
    if rk.rank != 0:
        return None
    proofs_per_step = job.nprov * job.batch
    n_local = args.steps * proofs_per_step
    n_proofs = n_local * (1 if job.sharded else rk.world)
    cs = c.cs
    region = ("image -> proof: Wnn::proof = witness program on the device + create_proof, every proof for another image "
              "(zg_prover_prove_images)" if job.images else
              "create_proof from advice columns " + ("uploaded per proof from host memory" if HOST_ADVICE else "resident in HBM (rounds 1-4's region)"))
    out = {
        "metric": f"create_proof proofs/hour, {c.model_name}",
        "value": n_proofs / dt * 3600.0, "unit": "proofs/hour", "n_gpus": rk.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if job.sharded else "weak", "vs_baseline": None, "dtype": "u32x8 (254-bit Montgomery integers)",
        "data": ("checked-in model + example_image_7.png + 63 seeded noise images, seeded SRS" if args.model != "large"
                 else "synthetic: seeded stand-in model (the file is absent from the reference), seeded images and SRS"),
        "config": {"workload": f"{region}; zero_g WnnCircuit of {c.model_name}, k={c.k} ({cs.n_advice} advice, {cs.n_fixed} fixed, "
                               f"{len(cs.gates)} gates, {len(cs.lookups)} lookups, {len(cs.perm_columns)} equality columns, degree {cs.degree()}), "
                               f"proof {len(job.streams[0].last[0])} B",
                   "class_scores": c.scores, "proofs_per_step": proofs_per_step,
                   "parallelism": (f"{rk.world} rank(s), commitments sharded by point range (one all-gather per phase and prover, "
                                   f"{'in-library RCCL' if job.exchange_kind == 'rccl' else 'host callback on ' + rk.backend}), "
                                   f"{job.nprov} prover(s)/rank x batches of {job.batch}" if job.sharded else
                                   f"{rk.world} GPU(s) x {job.nprov} prover stream(s) x lock-step batches of {job.batch} proofs")},
        "mode": args.mode if rk.world > 1 else "single-gpu",
        "timed_region": "image_to_proof" if job.images else "from_resident_columns",
        "exchange": job.exchange_kind, "rccl_ranks": job.rccl_ranks, "collective": collective_object(rk),
        "ranks_share_a_device": bool(rk.world > 1 and "ZG_BENCH_DEVICE" in os.environ),
        "inputs": "image bytes (784 B per proof over PCIe); SRS and proving key resident in HBM" if job.images else
                  ("host memory, uploaded per proof" if HOST_ADVICE else "resident in HBM"),
        "proofs_per_step": proofs_per_step, "ms_per_proof": dt / n_local * 1e3, "proof_len": len(job.streams[0].last[0]),
        "runtime_env": RUNTIME_ENV, "provers_per_gpu": job.nprov, "batch": job.batch,
        "launches_per_proof": sum(v[0] for v in stats.values()) / max(1, n_local),
        "algorithmic_bytes_per_proof": algorithmic_bytes_per_proof(cs),
        "algorithmic_GBps": algorithmic_bytes_per_proof(cs) / (dt / n_local) / 1e9,
        "detail_file": os.path.relpath(DETAIL, ROOT),
    }
    if LAST_POWER.get("cards"):
        out["power"] = dict(LAST_POWER)  # (this rank's card during the timed steps)
        # energy of one proof on this rank's card: what a power-limited chip is really short of
        if not (rk.world > 1 and "ZG_BENCH_DEVICE" in os.environ):  # (a rehearsal's ranks share one card: its power is not this rank's)
            out["power"]["joules_per_proof"] = LAST_POWER["cards"][0]["power_w_avg"] * dt / n_local
    if job.images:
        out["witness_program"] = c.witness_program()[1]
    if stats:
        pmc = load_pmc(args.model) if not job.sharded else None
        # (the counter passes were taken at pmc["proofs_per_launch"] proofs per launch: every kernel of the path processes
        #  the proofs of a batch side by side, so bytes per launch scale with the batch)
        job.pmc, job.pmc_scale = pmc, (job.batch / float(pmc.get("proofs_per_launch", job.batch)) if pmc else 1.0)
        roofline, kernels, device_ms, charged = roofline_object(stats, None, pmc, job.pmc_scale, n_local, job.batch)
        out.update({"roofline": roofline, "kernels": kernel_table(kernels, device_ms), "device_ms_per_proof": device_ms / n_local,
                    # SURVEY.md 8d's per-proof figure as the library charged it launch by launch: must equal the formula's
                    "algorithmic_bytes_per_proof_charged": charged / n_local, "valu": valu_object(pmc, dt / n_local * 1e3)})
    return out


def tail_legs(job, em: Emitter):
    """Everything after the headline, each leg in its own try (Emitter.leg): the check of the timed proofs, the same provers
    from resident columns (continuity with rounds 1-4), one prover alone (roofline.serialised), the lone-proof pairs, the
    other three models, the CPU baseline."""
    args, rk, c, out = job.args, job.rk, job.circuit, em.out
    single = rk.world == 1 and not job.sharded
    if not args.no_verify:
        em.leg("verify", lambda: verify_last_step(c, job.streams, host_cores()))
    if single and job.stats and not args.no_serialised:
        def serial_leg():
            serial = serialised_pass(job.streams, job.ctxs, job.barrier, job.batch, job.pmc, job.pmc_scale)
            n_local = args.steps * job.nprov * job.batch
            roofline, _, _, _ = roofline_object(job.stats, serial, job.pmc, job.pmc_scale, n_local, job.batch)
            return {"roofline": roofline}
        em.leg("serialised", serial_leg)
    if single and job.images and not args.no_from_resident:
        def resident_leg():
            for s in job.streams:
                s.disable_images()
            dt, _ = measure(job.streams, job.ctxs, 5, 1, job.barrier)
            return {"from_resident_columns_ms_per_proof": dt / (5 * job.nprov * job.batch) * 1e3}
        em.leg("from_resident", resident_leg)
    if job.lone is not None:
        def lone_leg():
            got = lone_probe(job.lone, tables=True, gate=os.environ.get("ZG_LAT_GATE", "1") != "0", images=job.images)
            return {"lone": {**out.get("lone", {}), "opted_in": got, "opted_in_is": "zg_prover_enable_digit_tables + ZG_LAT_GATE=1 + "
                             "bench.py's runtime_env; `default` = none of them, in a child process"},
                    "create_proof_wall_s": got["create_proof_ms"] / 1e3,
                    "image_to_proof_wall_s": got.get("image_to_proof_ms", 0.0) / 1e3 if "image_to_proof_ms" in got else None,
                    "lone_proof_digit_table_bytes": got["digit_table_bytes"], "lone_proof_gate": got["gate"],
                    "single_proof_phase_ms": got["phase_ms"]}
        em.leg("lone_opted_in", lone_leg)
        if not args.no_lone_default:
            em.leg("lone_default", lambda: {"lone": {**out.get("lone", {}), "default": lone_default_in_child(args.model, job.images)}})
    if single and args.model == "tiny" and not args.no_other_configs:
        # the other three models of BASELINE.json: a few steps each, same driver.  The headline's provers go first (and its
        # base sets: the lone probe left its digit tables on them, 3 x 26 GB, and k = 17 wants 12 x 8 slots of 1.4 GiB)
        close_streams(job.ctxs, job.streams, job.lone)
        job.streams, job.lone, job.ctxs = [], None, job.ctxs[:1]
        c.free_bases()
        out["other_configs"] = {}
        for m in ("small", "medium", "large"):
            em.leg(f"other_{m}", lambda m=m: other_config(job, m, out["other_configs"]))
    if rk.rank == 0 and not args.no_cpu_baseline and rk.world == 1:
        em.leg("cpu_baseline", lambda: {"cpu_baseline": cpu_baseline(c, out["proof_len"], host_cores())})


def other_config(job, m: str, into: dict):
    """One of the other BASELINE models on the same GPU: image -> proof steps (3 timed after 1 warm-up), one proof checked,
    one prover alone (its `roofline`), its lone pair (opted-in form)."""
    args, rk = job.args, job.rk
    c2 = Circuit(job.ctx0, m)
    b2, np2 = min(job.batch, DEFAULT_BATCH[m]), min(job.nprov, 12)
    cx, st2, lone2 = make_streams(rk.dev_index, c2, job.ctx0, np2, b2, rk.rank, lone=not args.no_latency_probe)
    rec = into.setdefault(m, {"model": c2.model_name, "k": c2.k, "batch": b2, "provers": np2})
    try:
        if job.images:
            arrays, summary = c2.witness_program()
            pool = image_pool(c2)
            for s in st2:
                s.enable_images(arrays, pool)
            rec["witness_program"] = summary

        def barrier():
            torch.cuda.synchronize(rk.dev)
            for x in cx:
                x.sync()

        steps = 3
        dt2, stats2 = measure(st2, cx, steps, 1, barrier, profile=not args.no_kernel_events)
        n2 = steps * np2 * b2
        rec.update({"ms_per_proof": dt2 / n2 * 1e3, "proofs_per_hour": n2 / dt2 * 3600.0,
                    "timed_region": "image_to_proof" if job.images else "from_resident_columns"})
        if LAST_POWER.get("cards"):
            rec["power"] = LAST_POWER["cards"]
        if not args.no_verify:
            rec.update(verify_last_step(c2, st2, host_cores(), byte_checks=1))
        if stats2 and not args.no_serialised:
            pmc2 = load_pmc(m)
            scale2 = b2 / float(pmc2.get("proofs_per_launch", b2)) if pmc2 else 1.0
            serial2 = serialised_pass(st2, cx, barrier, b2, pmc2, scale2, steps=2)
            rf, _, _, _ = roofline_object(stats2, serial2, pmc2, scale2, n2, b2)
            rec["roofline"] = {**_pick(rf, ("kernel", "achieved", "frac", "avg_launch_ms", "algo_bytes_per_launch", "msm_counter_over_algorithmic",
                                            "counter_file")), "traffic": rf.get("traffic"),
                               "serialised": _pick(rf["serialised"], ("frac", "avg_launch_ms", "ms_per_proof", "device_ms_per_proof")),
                               "serialised_kernels": dict(list(rf["serialised"]["kernels"].items())[:12]),
                               "families": {f: _pick(v, ("share_of_device_time", "algo_GBps", "counter_over_algorithmic_bytes"))
                                            for f, v in rf["families"].items()}}
            rec["valu"] = _pick(valu_object(pmc2, rec["ms_per_proof"]), ("wave_instructions_per_proof", "frac_of_four_cycle_issue_rate"))
        if lone2 is not None:
            got = lone_probe(lone2, tables=True, gate=os.environ.get("ZG_LAT_GATE", "1") != "0", images=job.images, repeats=5)
            rec.update({"create_proof_wall_s": got["create_proof_ms"] / 1e3,
                        "image_to_proof_wall_s": got["image_to_proof_ms"] / 1e3 if "image_to_proof_ms" in got else None, "lone": got})
    finally:
        close_streams(cx, st2, lone2)
        c2.free_bases()
    return None


def proofs_main(args, rk) -> int:
    job = build_job(args, rk)
    em = Emitter() if rk.rank == 0 else None
    out = measure_headline(job)
    if rk.rank == 0:
        em.early(out)
        tail_legs(job, em)
        em.final()
    if rk.dist is not None:
        rk.dist.barrier()  # (rank 0 checks its proofs against the oracle after the timed region: the others wait here)
        rk.dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------------ --mode msm-only

def fill_scalars(k: int, vectors: int, seed: int) -> np.ndarray:
    """SURVEY.md 8d's synthetic scalars, uint64[vectors, n, 4] in the Montgomery form: of every 30 vectors (one proof's
    commitments) the first 6 are "advice-like" (70 % zero, 20 % in {0, 1}, 8 % < 2^8, 2 % uniform), the rest uniform."""
    n = 1 << k
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 1 << 64, size=(vectors, n, 4), dtype=np.uint64)
    out[:, :, 3] &= np.uint64((1 << 61) - 1)  # (< 2^253 < r: a Montgomery representative of a uniform-looking scalar)
    mont = np.array([limbs(i * MONT % R) for i in range(256)], dtype=np.uint64)
    for v in range(vectors):
        if v % 30 < 6:
            u = rng.random(n)
            small = rng.integers(0, 256, size=n, dtype=np.int64)
            ints = np.where(u < 0.7, 0, np.where(u < 0.9, small & 1, small))
            out[v] = np.where((u >= 0.98)[:, None], out[v], mont[ints])
    return out


def msm_only_check(sums, scal, order, on_gl, g, gl, vectors: int) -> dict:
    """sampled sums of the LAST timed step against the oracle's best_multiexp over the WHOLE vectors (after the timed region)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc

    orc.load().orc_set_threads(host_cores())
    picks = sorted({0, 5, 6, 19, min(20, vectors - 1), min(29, vectors - 1), vectors - 1})
    for pos in picks:
        v = int(order[pos])
        want = orc.msm(scal[v], gl if on_gl[v] else g)
        if not np.array_equal(sums[pos], want):
            return {"verified": False, "detail": f"MSM {v} differs from the oracle's best_multiexp"}
    return {"verified": True, "detail": f"{len(picks)} of the last step's {vectors} sums == oracle best_multiexp over the whole vectors"}


def msm_only_cpu_baseline(scal, on_gl, g, gl, k: int) -> dict:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc

    orc.load().orc_set_threads(host_cores())
    orc.msm(scal[6], gl)  # warm-up
    ts = []
    for v in (6, 7, 20, 21, 0):
        t0 = time.perf_counter()
        orc.msm(scal[v], gl if on_gl[v] else g)
        ts.append(time.perf_counter() - t0)
    return {"cpu_baseline": {"value": 1.0 / median(ts), "unit": "MSMs/s", "cores": host_cores(), "kind": "port", "wall_s": median(ts),
                             "samples_s": [round(t, 5) for t in ts],
                             "sample": f"median of 5 oracle best_multiexp (Pippenger, OpenMP x{host_cores()}), n = 2^{k}"}}


def msm_only_main(args, rk) -> int:
    """north_star's scaling target by itself: B = batch x 30 commitment MSMs of the model's size (20 against g_lagrange, 10
    against g per proof), sharded by point range -- rank r multiplies scalars [lo_r, hi_r) of every vector against its
    resident slice of both base sets (ONE launch sequence per base set), ONE all-gather of the B partial sums (128 B each),
    rank-order EC additions and the normalisation on every rank.  value = MSMs / s.  After the timed region sampled sums
    are compared with the oracle's best_multiexp over the WHOLE vectors."""
    k, model_name = MODELS[args.model]
    n = 1 << k
    batch = max(1, args.batch if args.batch else DEFAULT_BATCH[args.model])
    vectors = batch * 30
    ctx = zg.Ctx(rk.dev_index)
    s = np.array(limbs(0x5EED5EED5EED5EED * MONT % R), dtype=np.uint64)
    g, gl = ctx.params_new(k, s)
    stub = args.stub_world if (args.stub_world > 1 and rk.world == 1) else 0
    world = stub or rk.world
    lo, hi = rk.rank * n // world, (rk.rank + 1) * n // world
    g_bases, gl_bases = ctx.register_bases(g[lo:hi]), ctx.register_bases(gl[lo:hi])
    # the form the prover multiplies random vectors in: free-position odd digits against one table row per bit position
    # (csrc/prover.hip, zg_prover_create: width 16 from 2^16 points, 15 from 2^14, else log2 + 1 -- by the SLICE's size)
    lg = max(1, (hi - lo).bit_length() - 1)
    width = 16 if lg >= 16 else 15 if lg >= 14 else max(3, lg + 1)
    ctx.enable_bit_table(g_bases, width)
    ctx.enable_bit_table(gl_bases, width)
    ctx.set_msm_latency(False)  # the THROUGHPUT form of the stand-alone entry, as a prover in that form multiplies (a context starts in the latency form)
    scal = fill_scalars(k, vectors, 77)
    on_gl = np.array([v % 30 < 20 for v in range(vectors)])
    order = np.concatenate([np.nonzero(on_gl)[0], np.nonzero(~on_gl)[0]])  # (g_lagrange vectors first: two launch sequences)
    n_gl = int(on_gl.sum())
    d_scal = torch.from_numpy(np.ascontiguousarray(scal[order][:, lo:hi, :]).view(np.int64)).to(rk.dev)
    d_part = torch.zeros(vectors * 16, dtype=torch.int64, device=rk.dev)  # XYZZ, 128 B each
    d_all = torch.zeros(world * vectors * 16, dtype=torch.int64, device=rk.dev)
    d_sum = torch.zeros(vectors * 16, dtype=torch.int64, device=rk.dev)
    stride, m = hi - lo, hi - lo
    torch.cuda.synchronize(rk.dev)  # (torch's stream made the buffers; the library works on its own)
    stream = torch.cuda.ExternalStream(ctx.stream, device=rk.dev)
    lib = ctx.lib
    import ctypes

    CHUNK = 240  # vectors per launch sequence: what one commitment phase of a batch of proofs carries (192-288)
    # the launch sequences of a step are spread over `streams` contexts (own HIP stream + workspace each, one host thread
    # each; the base tables are shared), as the provers of a GPU overlap the latency-bound reductions of one sequence with the
    # accumulation of another
    nstreams = max(1, args.provers if args.provers else 4)
    xs = [ctx] + [zg.Ctx(rk.dev_index) for _ in range(nstreams - 1)]
    for x in xs:
        x.set_msm_latency(False)
    work = [(bases, v0, min(CHUNK, first + count - v0)) for bases, first, count in ((gl_bases, 0, n_gl), (g_bases, n_gl, vectors - n_gl))
            for v0 in range(first, first + count, CHUNK)]

    def run_share(j):
        for bases, v0, nv in work[j::nstreams]:
            xs[j].msm_batch_dev(bases, d_scal.data_ptr() + v0 * stride * 32, stride, nv, m, d_part.data_ptr() + v0 * 128)
        xs[j].sync()

    def step():
        th = [threading.Thread(target=run_share, args=(j,)) for j in range(1, nstreams)]
        for t in th:
            t.start()
        run_share(0)
        for t in th:
            t.join()
        if rk.dist is not None or stub:
            if stub:  # (what the gather would deliver, made locally: G copies of this rank's partial sums, on the MSM's stream)
                with torch.cuda.stream(stream):
                    d_all.view(world, -1).copy_(d_part.unsqueeze(0).expand(world, -1))
            elif rk.backend == "nccl":
                with torch.cuda.stream(stream):  # (the collective on the MSM's own stream: no host round trip in between)
                    rk.dist.all_gather_into_tensor(d_all, d_part)
            else:
                ctx.sync()
                parts = [torch.empty(vectors * 16, dtype=torch.int64) for _ in range(rk.world)]
                rk.dist.all_gather(parts, d_part.cpu())
                d_all.copy_(torch.cat(parts))
                torch.cuda.synchronize(rk.dev)
            st = lib.zg_xyzz_sum_ranks_dev(ctx.h, ctypes.c_void_p(d_all.data_ptr()), ctypes.c_size_t(world), ctypes.c_size_t(vectors),
                                           ctypes.c_void_p(d_sum.data_ptr()))
            assert st == 0, zg.ZgError(st, "zg_xyzz_sum_ranks_dev")
            return ctx.msm_finish(d_sum.data_ptr(), vectors)
        return ctx.msm_finish(d_part.data_ptr(), vectors)

    def barrier():
        if rk.dist is not None:
            rk.dist.barrier()
        torch.cuda.synchronize(rk.dev)
        ctx.sync()

    if WATCHDOG:
        WATCHDOG.arm(True)
    for _ in range(max(1, args.warmup)):
        step()
    for x in xs:
        x.profile(not args.no_kernel_events)
    barrier()
    sampler = PowerSampler().start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sums = step()
        if WATCHDOG:
            WATCHDOG.pat()
    barrier()
    dt = max_over_ranks(rk, time.perf_counter() - t0)
    power = sampler.stop()
    if WATCHDOG:
        WATCHDOG.arm(False)
    stats = {}
    for x in xs:
        if not args.no_kernel_events:
            for name, (l, ms, by, ub) in x.profile_collect().items():
                a = stats.get(name, (0, 0.0, 0.0, 0.0))
                stats[name] = (a[0] + l, a[1] + ms, a[2] + by, a[3] + ub)
        x.profile(False)
    if rk.rank == 0:
        em = Emitter()
        unit = n * 96 + 96
        out = {"metric": f"KZG commitment MSMs/s (best_multiexp, n = 2^{k}), shapes of {model_name}",
               "value": args.steps * vectors / dt, "unit": "MSMs/s", "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "u32x8 (254-bit Montgomery integers)", "data": "synthetic: seeded scalars (SURVEY 8d mix), seeded SRS",
               "config": {"workload": f"{vectors} MSMs per step = {batch} proofs x 30 commitments (20 over g_lagrange, 10 over g; 6 advice-like, "
                                      f"24 uniform scalars) of n = 2^{k} points, sharded by point range: {hi - lo} points per rank, ONE all-gather "
                                      f"of {vectors} x 128 B partial sums per step, local EC additions + normalisation",
                          "proofs_per_step": batch, "parallelism": f"{world} rank(s) x point range n/{world}; {nstreams} streams per rank"},
               "mode": "msm-only", "msms_per_step": vectors, "digit_width": width, "points_per_rank": hi - lo, "stub_world": stub or None, "us_per_msm": dt / (args.steps * vectors) * 1e6,
               "collective": collective_object(rk), "ranks_share_a_device": bool(rk.world > 1 and "ZG_BENCH_DEVICE" in os.environ),
               "runtime_env": RUNTIME_ENV, "detail_file": os.path.relpath(DETAIL, ROOT)}
        if power.get("cards"):
            out["power"] = power
        if stats:
            kernels, families, device_ms, _ = roofline_tables(stats, None, 1.0)
            acc = kernels.get("msm_accumulate")
            if acc:
                per_launch = acc["unit_bytes_per_launch"] or acc["algo_bytes_per_launch"]
                ach = per_launch / (acc["avg_launch_ms"] * 1e-3) / 1e9
                out["roofline"] = {"bound": "hbm", "kernel": "msm_accumulate", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                   "frac": ach / HBM_PEAK_GBPS, "traffic": None, "avg_launch_ms": acc["avg_launch_ms"],
                                   "algo_bytes_per_launch": per_launch, "families": families,
                                   "whole_msm_GBps": args.steps * vectors * (hi - lo) / n * unit / (device_ms * 1e-3) / 1e9 if device_ms else None}
            out["kernels"] = kernel_table(kernels, device_ms)
            out["device_ms_per_step"] = device_ms / args.steps
        em.early(out)

        if stub:
            out["n_gpus"] = 1
            out["config"]["workload"] = (f"COMPUTE LEG of rank 0 of a world of {stub}, no collective, NOT a scaling figure (sums not checked: the "
                                         f"other ranks' partial sums are copies of this rank's); ") + out["config"]["workload"]
        if not args.no_verify and not stub:
            em.leg("verify", lambda: msm_only_check(sums, scal, order, on_gl, g, gl, vectors))
        if not args.no_cpu_baseline and rk.world == 1:
            em.leg("cpu_baseline", lambda: msm_only_cpu_baseline(scal, on_gl, g, gl, k))
        em.final()
    if rk.dist is not None:
        rk.dist.barrier()
        rk.dist.destroy_process_group()
    for x in xs[1:]:
        x.close()
    g_bases.free()
    gl_bases.free()
    ctx.close()
    return 0


# ------------------------------------------------------------------------------------------------ --dry-run (no GPU)

def dry_run_main(args, rk) -> int:
    """The N-rank plumbing without a GPU (CPU tests; the 8-rank shape the driver launches cannot be rehearsed on a one-GPU
    box, which allows six GPU processes): launcher -> torch.distributed.run -> gloo group -> ranks counted by the backend
    -> one exchange group per prover (shard-msm's bring-up order) -> barriers -> MAX over ranks -> ONE line from rank 0.
    No proof is made and the line says so (`dry_run`, value 0)."""
    dist = rk.dist
    nprov = max(1, args.provers or 1)
    if dist is not None and args.mode == "shard-msm":
        with stdout_to_stderr():
            groups = [dist.new_group(backend=None) for _ in range(nprov)]
            for g in groups:
                dist.barrier(group=g)
            for g in groups:  # one all-gather per group, as a commitment phase does
                parts = [torch.empty(16, dtype=torch.uint8) for _ in range(rk.world)]
                dist.all_gather(parts, torch.full((16,), rk.rank, dtype=torch.uint8), group=g)
                assert [int(p[0]) for p in parts] == list(range(rk.world))
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + rk.rank))
    if dist is not None:
        dist.barrier()
    dt = max_over_ranks(rk, time.perf_counter() - t0)
    if rk.rank == 0:
        out = {"metric": "DRY RUN (no GPU work): rank plumbing only", "value": 0.0, "unit": "proofs/hour", "n_gpus": rk.world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
               "scaling": "strong" if args.mode != "replicas" else "weak", "vs_baseline": None, "dtype": "none", "data": "none (dry run)",
               "config": {"workload": "no proof is made: launcher, process group, per-prover exchange groups, barriers, max over ranks",
                          "proofs_per_step": 0, "parallelism": f"{rk.world} rank(s), gloo, no GPU"},
               "mode": args.mode, "dry_run": True, "collective": collective_object(rk)}
        print(format_line(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------------ main

def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", choices=sorted(MODELS), default="tiny",
                    help="tiny = model_28input_256entry_1hash_1bpi (k=14, the BASELINE metric's configuration)")
    ap.add_argument("--batch", type=int, default=None, help="proofs per lock-step batch (default 32 / 16 / 16 / 8 by model)")
    ap.add_argument("--provers", type=int, default=None,
                    help="proof streams per GPU (provers sharing one proving key): 12; in shard-msm 4 with the host exchange "
                         "(an exchange group per prover) and 1 with raw RCCL communicators")
    ap.add_argument("--mode", choices=["replicas", "shard-msm", "msm-only"], default="replicas")
    ap.add_argument("--exchange", choices=["host", "rccl"], default=None,
                    help="shard-msm: all-gather inside the library on a raw RCCL communicator per prover "
                         "(zg_prover_set_shard_rccl; the default when every rank owns a GPU: ONE prover per rank then) or "
                         "through a host callback on torch.distributed groups, one per prover (the default of one-GPU "
                         "rehearsals: RCCL refuses two ranks on one device)")
    ap.add_argument("--from-resident", "--no-image-to-proof", dest="from_resident", action="store_true",
                    help="time create_proof from advice columns resident in HBM (rounds 1-4's region) instead of image -> proof")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="no per-launch HIP events in the timed region (no roofline object)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of the other three models")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--host-advice", action="store_true",
                    help="every proof uploads its advice columns from host memory (PCIe-inclusive rate, for DESIGN.md)")
    ap.add_argument("--no-serialised", action="store_true", help="skip the one-prover pass behind roofline.serialised")
    ap.add_argument("--no-from-resident", action="store_true", help="skip the from-resident-columns pass kept for continuity")
    ap.add_argument("--no-latency-probe", action="store_true", help="skip the lone-proof latency measurements (counter passes)")
    ap.add_argument("--no-lone-default", action="store_true", help="skip the child process behind lone.default")
    ap.add_argument("--tail-only-headline", action="store_true", help="no tail legs at all: the timed region and the line")
    ap.add_argument("--stub-world", type=int, default=0,
                    help="msm-only on ONE GPU: this process is rank 0 of a world of G -- points [0, n / G) of both base sets, the other "
                         "ranks' partial sums replaced by copies of its own (no collective, sums not checked): one rank's COMPUTE LEG, "
                         "never a scaling figure")
    ap.add_argument("--lone-child", choices=sorted(MODELS), default=None, help=argparse.SUPPRESS)
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    if args.tail_only_headline:
        args.no_cpu_baseline = args.no_other_configs = args.no_verify = args.no_serialised = True
        args.no_from_resident = args.no_latency_probe = args.no_lone_default = True
    return args


def main():
    global HOST_ADVICE
    args = parse_args()
    HOST_ADVICE = args.host_advice
    # --gpus N > 1 outside torch.distributed.run: start the N ranks as a child and relay its line -- decided BEFORE the
    # first GPU call of this process (setup_ranks' torch.cuda.is_available() initialises the runtime)
    cmd = launcher_command(sys.argv[1:], args.gpus, os.environ)
    if cmd is not None:
        raise SystemExit(relay(cmd))
    if args.lone_child:
        raise SystemExit(lone_child(args))
    rk = setup_ranks(args)
    if args.dry_run:
        raise SystemExit(dry_run_main(args, rk))
    if args.mode == "msm-only":
        raise SystemExit(msm_only_main(args, rk))
    raise SystemExit(proofs_main(args, rk))


if __name__ == "__main__":
    main()
