#!/bin/bash
# usage: tools/ab_model.sh MODEL "LABEL ENV=VAL ..." ...   (short runs of one model, 4 provers)
model=$1; shift
out=gpurun_out/ab_model_$model.txt
: > $out
for cfg in "$@"; do
  label=${cfg%% *}
  envs=${cfg#* }
  [ "$envs" = "$cfg" ] && envs=""
  env $envs python bench.py --model $model --provers 4 --batch ${ZG_AB_BATCH:-16} --steps 4 --warmup 1 --no-other-configs --no-cpu-baseline --no-verify --no-latency-probe 2>/dev/null \
    | python -c "import json,sys; d=json.load(sys.stdin); print('$model $label ms/proof %.4f' % d['ms_per_proof'])" >> $out
done
cat $out
