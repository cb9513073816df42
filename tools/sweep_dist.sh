#!/bin/bash
# throughput with the RCCL control plane initialised (one rank), versus hardware queue count
for q in 16 18 20 24; do
  GPU_MAX_HW_QUEUES=$q ZG_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > /tmp/d.json
  python - $q <<'PY'
import json,sys
d=json.load(open('/tmp/d.json')); print("rccl hwq", sys.argv[1], round(d["ms_per_proof"],3), "ms/proof", flush=True)
PY
done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > /tmp/d.json; python -c "
import json; d=json.load(open('/tmp/d.json')); print('no rccl', round(d['ms_per_proof'],3))"
