#!/bin/bash
# window-size sensitivity of the MSM inside the full proof (tuning aid): ./tools/sweep_c.sh MODEL C...
M=${1:-tiny}; shift
for c in "$@"; do
  ZG_MSM_C=$c python bench.py --model $M --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null > /tmp/b.json
  python - "$c" <<'PY'
import json,sys
d=json.load(open('/tmp/b.json')); k=d["single_proof_kernels_ms"]
print("c", sys.argv[1], round(d["ms_per_proof"],3), "ms/proof, latency", round(d["create_proof_wall_s"]*1e3,2), {n:k.get(n) for n in ("msm_accumulate","msm_bucket_scan","msm_bucket_sum","msm_scan","msm_heavy")})
PY
done
