#!/bin/bash
# A/B of a lone proof (latency form) on the GPU box: each argument "LABEL ENV=VAL ..." runs tools/lone_proof.py MODEL
model=${MODEL:-tiny}
out=gpurun_out/ab_lone.txt
: > $out
for cfg in "$@"; do
  label=${cfg%% *}
  envs=${cfg#* }
  [ "$envs" = "$cfg" ] && envs=""
  for rep in 1 2; do
    echo "$label $(env $envs python tools/lone_proof.py $model 2>/dev/null | cut -c1-40)" >> $out
  done
done
cat $out
