#!/bin/bash
# hardware-queue / stream-count sensitivity of the throughput bench (tuning aid)
for q in 16 24 32; do for s in 8 12 16; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 8 --warmup 1 --streams $s --no-cpu-baseline 2>/dev/null > /tmp/c.json
  python - "$q" <<'PY'
import json,sys
d=json.load(open('/tmp/c.json'))
print("hwq", sys.argv[1], "streams", d["streams_per_gpu"], round(d["ms_per_proof"],3), "ms/proof", round(d["value"]), "proofs/h, latency", round(d["create_proof_wall_s"]*1e3,2))
PY
done; done
