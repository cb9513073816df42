#!/usr/bin/env python3
"""A few lone proofs of the headline circuit in the latency form (for rocprofv3 --kernel-trace + tools/timeline.py)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (sets the paths and GPU_MAX_HW_QUEUES)

zg = bench.zg
ctx = zg.Ctx(0)
c = bench.Circuit(ctx, sys.argv[1] if len(sys.argv) > 1 else "tiny")
p = zg.Prover(ctx, c.img, c.fixed, c.sigma, c.g_bases, c.gl_bases, c.vk_repr)
p.set_overlap(True)
for i in range(4):
    p.prove(c.advice, c.instance, i)
t0 = time.perf_counter()
for i in range(4):
    p.prove_dev(p.advice_slot(0), c.instance, 10 + i)
print("lone proof ms", (time.perf_counter() - t0) / 4 * 1e3, p.phase_ms())
p.close()
