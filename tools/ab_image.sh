#!/bin/bash
# A/B of the image_to_proof leg (a different image per proof): each argument "LABEL ENV=VAL ..."
out=gpurun_out/ab_image.txt
: > $out
for cfg in "$@"; do
  label=${cfg%% *}
  envs=${cfg#* }
  [ "$envs" = "$cfg" ] && envs=""
  env $envs python bench.py --steps 6 --warmup 2 --no-other-configs --no-cpu-baseline --no-verify --no-latency-probe 2>/dev/null \
    | python -c "import json,sys; d=json.load(sys.stdin); print('$label same-image %.4f image_to_proof %.4f' % (d['ms_per_proof'], d['image_to_proof']['ms_per_proof']))" >> $out
done
cat $out
