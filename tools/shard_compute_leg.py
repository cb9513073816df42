#!/usr/bin/env python3
"""ONE rank's compute leg of bench.py's point-range-sharded step, measured on one GPU.

Rank 0 of a world of G holds points [0, n / G) of ParamsKZG::g and ::g_lagrange (zg_prover_set_shard), proves the bench's
step -- `--provers` forked provers x lock-step batches -- and its exchange is a STUB (tools/shard_stub.c: the other ranks'
partial sums are the identity, nothing is communicated).  What is timed is therefore exactly what one GPU of G computes per
proof: its slice of every commitment MSM, and ALL of the transforms, evaluate_h, products and openings, which north_star
keeps per GPU.  It is a projected compute leg with no collective -- NOT a scaling curve: the all-gather of G x commitments x
128 B per phase (five per batch) and its latency over xGMI come on top, and no multi-GPU node was available to measure them.

    python tools/shard_compute_leg.py [--models tiny medium large] [--gpus 1 2 4 8] [--provers 12] > profiles/r04/shard_compute_leg.json
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (paths, GPU_MAX_HW_QUEUES, torch before the library)

zg = bench.zg


class CStub:
    """the stub exchange as a C function pointer (no Python, no GIL inside the provers' threads)"""

    def __init__(self, world: int):
        so = os.path.join(ROOT, "gpurun_out", "libshard_stub.so")
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "shard_stub.c"), "-o", so], check=True)
        self.lib = ctypes.CDLL(so)
        self.c_fn = ctypes.cast(self.lib.zg_stub_exchange, ctypes.c_void_p).value
        # user block: world, then the filler point of "rank 1" -- the generator (1, 2) with zz = zzz = 1, Montgomery form
        q = zg.FQ_MODULUS
        mont = lambda v: (v << 256) % q
        self.block = (ctypes.c_uint8 * 136)()
        raw = int(world).to_bytes(8, "little") + b"".join(mont(v).to_bytes(32, "little") for v in (1, 2, 1, 1))
        ctypes.memmove(self.block, raw, 136)
        self.c_user = ctypes.addressof(self.block)


def run(model: str, worlds, nprov: int, steps: int):
    ctx0 = zg.Ctx(0)
    c = bench.Circuit(ctx0, model)  # (whole base sets registered once: freed below, re-registered per slice)
    c.g_bases.free()
    c.gl_bases.free()
    n = 1 << c.k
    batch = {"tiny": 32, "small": 16, "medium": 16, "large": 8}[model]
    rows = []
    for G in worlds:
        c.lo, c.hi = 0, n // G
        c.g_bases = ctx0.register_bases(c.g[c.lo:c.hi])
        c.gl_bases = ctx0.register_bases(c.gl[c.lo:c.hi])
        stub = CStub(G)
        ctxs, streams, _ = bench.make_streams(0, c, ctx0, nprov, batch, 0, [stub] * nprov if G > 1 else None, (0, G))

        def barrier():
            for x in ctxs:
                x.sync()

        dt, stats = bench.measure(streams, ctxs, steps, 1, barrier, profile=True)
        proofs = steps * nprov * batch
        kernels, families, device_ms, _ = bench.roofline_tables(stats, None, 1.0)
        row = {"model": c.model_name, "k": c.k, "world": G, "points_per_rank": c.hi - c.lo, "provers": nprov, "batch": batch,
               "ms_per_proof": dt / proofs * 1e3, "device_ms_per_proof": device_ms / proofs,
               "family_device_ms_per_proof": {f: round(v["share_of_device_time"] * device_ms / proofs, 4) for f, v in families.items()},
               "msm_kernels_device_ms_per_proof": {k_: round(v["total_ms"] / proofs, 4) for k_, v in sorted(kernels.items(), key=lambda kv: -kv[1]["total_ms"])
                                                  if k_.startswith("msm_")}}
        rows.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
        for s in streams:
            s.prover.close()
        for x in ctxs[1:]:
            x.close()
        c.g_bases.free()
        c.gl_bases.free()
    ctx0.close()
    base = rows[0]
    for r in rows:
        r["speedup_vs_world_1"] = base["ms_per_proof"] / r["ms_per_proof"]
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", nargs="+", default=["tiny", "medium", "large"])
    ap.add_argument("--gpus", nargs="+", type=int, default=[1, 2, 4, 8])
    ap.add_argument("--provers", type=int, default=12)
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    t0 = time.time()
    out = {"_note": "PROJECTED COMPUTE LEG of one rank of a point-range-sharded proof, no collective, NOT a scaling curve: rank 0's slice "
                    "[0, n / world) of both base sets, stub exchange (the other ranks' partial sums = identity), every other kernel "
                    "of create_proof repeated in full as north_star prescribes; shared-chip device times (provers overlap), "
                    "ms_per_proof = wall time / proofs.  tools/shard_compute_leg.py",
           "rows": []}
    for m in args.models:
        out["rows"] += run(m, args.gpus, args.provers, args.steps)
    out["wall_s"] = round(time.time() - t0, 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
