#!/usr/bin/env python3
"""A few lone proofs of a model's circuit in both scheduling forms (for rocprofv3 --kernel-trace + tools/timeline.py):
    python tools/lone_proof.py [tiny|small|medium|large] [latency|throughput|both]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (sets the paths and GPU_MAX_HW_QUEUES)

zg = bench.zg
ctx = zg.Ctx(0)
c = bench.Circuit(ctx, sys.argv[1] if len(sys.argv) > 1 else "tiny")
which = sys.argv[2] if len(sys.argv) > 2 else "both"
p = zg.Prover(ctx, c.img, c.fixed, c.sigma, c.g_bases, c.gl_bases, c.vk_repr)
for form, overlap in (("latency", "tables"), ("throughput", False)):  # (ZG_LAT_FULL_C=0: the bucket form)
    if which not in (form, "both"):
        continue
    p.set_overlap(overlap)
    for i in range(3):
        p.prove(c.advice, c.instance, i)
    each = []
    for i in range(5):
        t0 = time.perf_counter()
        p.prove_dev(p.advice_slot(0), c.instance, 10 + i)
        each.append((time.perf_counter() - t0) * 1e3)
    print(f"{c.model} k={c.k} lone proof, {form} form: {sorted(each)[2]:.3f} ms (median of 5: {[round(x, 2) for x in each]})",
          [round(x, 3) for x in p.phase_ms()])
p.close()
