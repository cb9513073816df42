#!/usr/bin/env python3
"""witness_run alone on the chip, one image and a batch of 16, per model and ZG_WITNESS_LDS form (per-launch HIP events):
    python tools/witness_time.py [tiny small medium large]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ZG_BENCH_PLAIN_ENV", "1")
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

zg = bench.zg
ctx = zg.Ctx(0)
rows = []
for m in sys.argv[1:] or ["tiny", "small", "medium", "large"]:
    import witness_tape
    import wnn_model

    k, name = bench.MODELS[m]
    wnn = wnn_model.synthetic_wnn() if m == "large" else wnn_model.load_checked_in(name)
    arrays = witness_tape.trace(wnn, k).arrays()
    pool = np.stack([wnn_model.load_test_image().reshape(-1)] * 16)
    n = 1 << k
    bufs = [torch.zeros(6 * n * 4, dtype=torch.int64, device="cuda") for _ in range(16)]
    for form in (0, -1):
        zg.tuning_set("ZG_WITNESS_LDS", form)
        plan = zg.WitnessPlan(ctx, arrays)
        row = {"model": m, "k": k, "ZG_WITNESS_LDS": form, **plan.info()}
        for count in (1, 16):
            ptrs = [b.data_ptr() for b in bufs[:count]]
            for _ in range(3):
                plan.run(pool[:count], ptrs)
            ctx.profile(True)
            for _ in range(10):
                plan.run(pool[:count], ptrs)
            st = ctx.profile_collect()
            ctx.profile(False)
            row[f"witness_run_us_{count}"] = round(st["witness_run"][1] / st["witness_run"][0] * 1e3, 1)
            row[f"witness_finish_us_{count}"] = round(st["witness_finish"][1] / st["witness_finish"][0] * 1e3, 1)
        plan.close()
        rows.append(row)
        print(json.dumps(row), flush=True)
zg.tuning_set("ZG_WITNESS_LDS", -1)
