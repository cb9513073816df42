#!/bin/bash
# ms/proof over (provers, batch[, GPU_MAX_HW_QUEUES]): python bench.py in its lock-step configuration, headline model only
# usage: tools/sweep_batch.sh "P B [Q]" "P B [Q]" ...
out=gpurun_out/sweep_batch.txt
: > $out
for cfg in "$@"; do
  set -- $cfg
  q=${3:-16}
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 8 --warmup 2 --provers $1 --batch $2 --no-other-configs --no-cpu-baseline --no-verify --no-latency-probe --no-image-to-proof --no-serialised 2>/dev/null \
    | python -c "import json,sys; d=json.load(sys.stdin); print('provers $1 batch $2 queues $q ms/proof %.4f  device_ms/proof %.3f launches/proof %.2f' % (d['ms_per_proof'], d['device_ms_per_proof'], d['launches_per_proof']))" >> $out
done
cat $out
