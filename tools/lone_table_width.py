#!/usr/bin/env python3
"""A lone proof over digit tables of a given window width (ZG_LAT_FULL_C) and budget:
    python tools/lone_table_width.py MODEL C MAX_GB        (one process per setting: the tables belong to the base sets)"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

model, c_bits, max_gb = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
zg = bench.zg
if c_bits > 0:
    zg.tuning_set("ZG_LAT_FULL_C", c_bits)
ctx = zg.Ctx(0)
c = bench.Circuit(ctx, model)
p = zg.Prover(ctx, c.img, c.fixed, c.sigma, c.g_bases, c.gl_bases, c.vk_repr)
p.set_overlap(True)
t0 = time.perf_counter()
built = p.enable_digit_tables(int(max_gb * 1e9))
build_s = time.perf_counter() - t0
zg.tuning_set("ZG_LAT_GATE", 1)
for i in range(4):
    p.prove(c.advice, c.instance, i)
lat = []
for i in range(30):
    t0 = time.perf_counter()
    p.prove_dev(p.advice_slot(0), c.instance, 10 + i)
    lat.append((time.perf_counter() - t0) * 1e3)
print(f"{model} ZG_LAT_FULL_C={c_bits} budget {max_gb} GB: tables {built / 1e9:.1f} GB built in {build_s:.2f} s; lone proof median {statistics.median(lat):.4f} ms "
      f"(min {min(lat):.3f}); phases {[round(x, 3) for x in p.phase_ms()[:7]]}")
