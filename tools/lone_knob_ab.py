#!/usr/bin/env python3
"""Lone proofs of one prover under several settings of ONE launch-shape knob, interleaved in one process on one box:
    python tools/lone_knob_ab.py MODEL KNOB V1 V2 ... [--rounds R]        (value -1 = the library's default)
Proof bytes must not change; prints each setting's median latency.  Run under `timeout -k 10 SECONDS`."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (sets the paths and GPU_MAX_HW_QUEUES)

args = sys.argv[1:]
rounds = 5
if "--rounds" in args:
    i = args.index("--rounds")
    rounds = int(args[i + 1])
    del args[i:i + 2]
model, name, values = args[0], args[1], [int(v) for v in args[2:]]
zg = bench.zg
ctx = zg.Ctx(0)
c = bench.Circuit(ctx, model)
p = zg.Prover(ctx, c.img, c.fixed, c.sigma, c.g_bases, c.gl_bases, c.vk_repr)
p.set_overlap("tables")
want = {}
lat = {v: [] for v in values}
for r in range(rounds + 1):  # (round 0 warms every setting up)
    for v in values:
        zg.tuning_set(name, v)
        for i in range(10):
            t0 = time.perf_counter()
            proof = p.prove_dev(p.advice_slot(0), c.instance, 100 + i) if r else p.prove(c.advice, c.instance, 100 + i)
            if r:
                lat[v].append((time.perf_counter() - t0) * 1e3)
            if want.setdefault(i, proof) != proof:
                raise SystemExit(f"{name}={v}: proof {i} differs")
zg.tuning_set(name, -1)
print(f"{c.model} k={c.k} lone proof, {name}: " + ", ".join(f"{v}: {statistics.median(lat[v]):.4f} ms" for v in values)
      + f" (medians of {rounds * 10}; same proof bytes)")
p.close()
