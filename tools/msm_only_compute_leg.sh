#!/bin/bash
# One rank's compute leg of --mode msm-only at G = 1, 2, 4, 8 on ONE GPU (no collective: NOT a scaling curve) for three models;
# writes profiles/r05/msm_only_compute_leg.json.   ./tools/msm_only_compute_leg.sh   (GPU box)
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5/msm_only_leg
mkdir -p $OUT
for m in tiny medium large; do
  for g in 1 2 4 8; do
    python3 bench.py --mode msm-only --model $m --stub-world $g --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${m}_$g.json 2> $OUT/${m}_$g.err
    cp bench_detail.json $OUT/${m}_${g}_detail.json
    echo "$m G=$g done"
  done
done
python3 - <<EOF  # (on the GPU box this writes into the scratch copy: run the same lines here to file it under profiles/)
import json
rows = []
for m in ("tiny", "medium", "large"):
    base = None
    for g in (1, 2, 4, 8):
        d = json.load(open(f"$OUT/{m}_{g}_detail.json"))
        us = d["us_per_msm"]
        base = base or us
        k = d["kernels"]
        rows.append({"model": m, "world": g, "points_per_rank": d["points_per_rank"], "msms_per_step": d["msms_per_step"],
                     "digit_width": d["digit_width"], "us_per_msm": us, "x_vs_world_1": base / us, "msms_per_s_one_rank": d["value"],
                     "device_ms_per_step": d.get("device_ms_per_step"),
                     "kernel_ms_per_step": {n: round(v["avg_launch_ms"] * v_l, 4) for n, v in k.items()
                                            for v_l in [round(v["share_of_device_time"] * d["device_ms_per_step"] / max(v["avg_launch_ms"], 1e-9))]}})
json.dump({"_note": "bench.py --mode msm-only --stub-world G on ONE GPU: rank 0's COMPUTE LEG of the point-range-sharded commitment MSMs "
                    "(batch x 30 vectors of the model's size against points [0, n / G) of both base sets, the gather replaced by G local copies of "
                    "its own partial sums, the G-way EC additions and the normalisation real).  No collective, no other rank: NOT a scaling "
                    "curve -- what ONE GPU of G computes per MSM, before the all-gather of batch x 30 x 128 B per step over xGMI.",
           "rows": rows}, open("profiles/r05/msm_only_compute_leg.json", "w"), indent=1)
for r in rows:
    print(r["model"], r["world"], r["points_per_rank"], round(r["us_per_msm"], 2), round(r["x_vs_world_1"], 2))
EOF
