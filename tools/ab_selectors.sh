#!/bin/bash
# A/B inside the full bench: merged selector columns (halo2's keygen, default) vs one column per selector.
for r in $(seq 1 ${1:-2}); do for v in 0 1; do
  ZG_BENCH_NO_SELECTOR_COMPRESSION=$v python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null > /tmp/ab.json
  python - "$v" <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json')); k=d["single_proof_kernels_ms"]
print("uncompressed" if sys.argv[1]=="1" else "compressed  ", round(d["ms_per_proof"],3), "ms/proof  latency", round(d["create_proof_wall_s"]*1e3,2), {n:k.get(n) for n in ("evaluate_h","eval_batch","gwc_lincomb")}, flush=True)
PY
done; done
