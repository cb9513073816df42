#!/bin/bash
# window-size sensitivity of the MSM inside the full proof (tuning aid)
for c in 11 12 13 14 15 16; do
  ZG_MSM_C=$c python bench.py --steps 24 --warmup 4 --streams 4 --no-cpu-baseline 2>/dev/null > /tmp/b.json
  python - "$c" <<'PY'
import json,sys
d=json.load(open('/tmp/b.json')); k=d["single_proof_kernels_ms"]
print(sys.argv[1], round(d["ms_per_step"],3), round(d["create_proof_wall_s"]*1e3,2), {n:k.get(n) for n in ("msm_accumulate","msm_reduce1","msm_reduce2","msm_final","msm_count")})
PY
done
