#!/bin/bash
# bench line of every BASELINE configuration that fits one GPU: ./tools/bench_models.sh [models...]
for m in ${@:-tiny small medium large}; do
  s=16; [ $m = large ] && s=8
  python bench.py --model $m --streams $s --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null > /tmp/bm.json
  python - $m <<'PY'
import json,sys
d=json.load(open('/tmp/bm.json'))
print(sys.argv[1], "k:", d["config"]["workload"].split("k=")[1].split(",")[0], "ms/proof", round(d["ms_per_proof"],3), "proofs/h", round(d["value"]), "single-proof ms", round(d["create_proof_wall_s"]*1e3,2), "streams", d["streams_per_gpu"], flush=True)
PY
done
