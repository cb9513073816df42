#!/bin/bash
# A/B of bench variants on the GPU box: each line "LABEL ENV=VAL ..." runs the default bench with that environment
out=gpurun_out/ab_bench.txt
: > $out
for cfg in "$@"; do
  label=${cfg%% *}
  envs=${cfg#* }
  [ "$envs" = "$cfg" ] && envs=""
  for rep in 1 2; do
    env $envs python bench.py --steps ${STEPS:-10} --warmup 2 --no-other-configs --no-cpu-baseline --no-verify --no-latency-probe --no-image-to-proof 2>/dev/null \
      | python -c "import json,sys; d=json.load(sys.stdin); print('$label ms/proof %.4f device_ms/proof %.3f' % (d['ms_per_proof'], d['device_ms_per_proof']))" >> $out
  done
done
cat $out
