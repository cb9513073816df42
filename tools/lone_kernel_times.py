#!/usr/bin/env python3
"""Per-kernel device time of a LONE proof (latency form, digit tables) from the library's own launch events, under each of
several values of ZG_LAT_FULL_K (summands per task of the digit-table sums):
    python tools/lone_kernel_times.py MODEL K1 K2 ...
(ZG_LAT_FULL_C=c in the environment picks the tables' window width: one process per width.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

zg = bench.zg
ctx = zg.Ctx(0)
c = bench.Circuit(ctx, sys.argv[1])
p = zg.Prover(ctx, c.img, c.fixed, c.sigma, c.g_bases, c.gl_bases, c.vk_repr)
p.set_overlap("tables")
for i in range(3):
    p.prove(c.advice, c.instance, i)
for K in [int(x) for x in sys.argv[2:]]:
    zg.tuning_set("ZG_LAT_FULL_K", K)
    for i in range(2):
        p.prove_dev(p.advice_slot(0), c.instance, i)
    ctx.profile(True)
    t0 = time.perf_counter()
    for i in range(10):
        p.prove_dev(p.advice_slot(0), c.instance, i)
    dt = (time.perf_counter() - t0) / 10 * 1e3
    st = ctx.profile_collect()
    ctx.profile(False)

    def f(n):
        return (st[n][0] / 10, st[n][1] / 10 * 1e3) if n in st else (0, 0)

    print(f"K={K}: proof {dt:.3f} ms (with launch events); msm_accumulate_full {f('msm_accumulate_full')[1]:.0f} us in "
          f"{f('msm_accumulate_full')[0]:.0f} launches, msm_tree {f('msm_tree')[1]:.0f} us in {f('msm_tree')[0]:.0f}, "
          f"msm_digits {f('msm_digits')[1]:.0f} us", flush=True)
zg.tuning_set("ZG_LAT_FULL_K", -1)
p.close()
