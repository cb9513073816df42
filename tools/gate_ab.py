#!/usr/bin/env python3
"""ZG_LAT_GATE off against on for lone proofs of one prover, interleaved in ONE process on one box:
    python tools/gate_ab.py [tiny|small|medium|large] [rounds]
Checks that both settings give the same proof bytes for the same key, then prints the median latency of each.
Run it under `timeout -k 10 SECONDS`: a gate that never opened would leave the stream waiting."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (sets the paths and GPU_MAX_HW_QUEUES)

zg = bench.zg
ctx = zg.Ctx(0)
c = bench.Circuit(ctx, sys.argv[1] if len(sys.argv) > 1 else "tiny")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
p = zg.Prover(ctx, c.img, c.fixed, c.sigma, c.g_bases, c.gl_bases, c.vk_repr)
p.set_overlap("tables")
p.prove(c.advice, c.instance, 0)
print("warm", flush=True)
want = {}
lat = {0: [], 1: []}
for r in range(rounds):
    for gate in (0, 1):
        zg.tuning_set("ZG_LAT_GATE", gate)
        for i in range(10):
            t0 = time.perf_counter()
            proof = p.prove_dev(p.advice_slot(0), c.instance, 100 + i)
            lat[gate].append((time.perf_counter() - t0) * 1e3)
            if want.setdefault(i, proof) != proof:
                raise SystemExit(f"gate {gate}: proof {i} differs")
    print(f"round {r}: off {statistics.median(lat[0][-10:]):.3f} ms, on {statistics.median(lat[1][-10:]):.3f} ms", flush=True)
zg.tuning_set("ZG_LAT_GATE", -1)
print(f"{c.model} k={c.k}: gate off {statistics.median(lat[0]):.4f} ms, gate on {statistics.median(lat[1]):.4f} ms "
      f"(medians of {len(lat[0])}; same proof bytes)")
p.close()
