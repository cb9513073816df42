#!/bin/bash
for side in 1 0; do for s in 8 12 16 24; do
  ZG_SIDE_STREAM=$side python bench.py --steps 192 --warmup 24 --streams $s --no-cpu-baseline 2>/dev/null > /tmp/c.json
  python - $side <<'PY'
import json,sys
d=json.load(open('/tmp/c.json'))
print("side", sys.argv[1], "streams", d["streams_per_gpu"], round(d["ms_per_step"],3), "ms/proof", round(d["value"]), "proofs/h, latency", round(d["create_proof_wall_s"]*1e3,2))
PY
done; done
