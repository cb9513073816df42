#!/bin/bash
# A/B two builds of the library inside the full bench: ./tools/ab.sh A.so B.so [runs]
L=0g-halo2_amd/libzg_halo2.so
for r in $(seq 1 ${3:-3}); do for v in "$1" "$2"; do
  cp "$v" $L
  python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null > /tmp/ab.json
  python - "$v" <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json')); k=d["single_proof_kernels_ms"]
print(sys.argv[1].split('/')[-1], round(d["ms_per_proof"],3), "ms/proof  latency", round(d["create_proof_wall_s"]*1e3,2), {n:k.get(n) for n in ("msm_accumulate","ntt_rows","ntt_cols","evaluate_h","msm_bucket_scan")}, flush=True)
PY
done; done
