#!/bin/bash
# usage: tools/ab_model.sh MODEL "LABEL PROVERS BATCH [ENV=VAL ...]" ...   (short runs of one model)
model=$1; shift
out=gpurun_out/ab_model_$model.txt
: > $out
for cfg in "$@"; do
  set -- $cfg
  label=$1; p=$2; b=$3; shift 3
  env "$@" python bench.py --model $model --provers $p --batch $b --steps 4 --warmup 1 --no-other-configs --no-cpu-baseline --no-verify --no-latency-probe --no-image-to-proof 2>/dev/null \
    | python -c "import json,sys; d=json.load(sys.stdin); print('$model $label provers $p batch $b ms/proof %.4f' % d['ms_per_proof'])" >> $out
done
cat $out
