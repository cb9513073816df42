#!/bin/bash
# Same-box A/B of the throughput bench between ENVIRONMENTS (knobs), alternating: REPS rounds over all the configurations.
#   tools/ab_env.sh "base" "aff2 ZG_MSM_AFFINE=2" ...        -> gpurun_out/ab_env.txt (one line per run) and a summary
out=gpurun_out/${OUT:-ab_env}.txt
: > $out
for rep in $(seq 1 ${REPS:-3}); do for cfg in "$@"; do
  label=${cfg%% *}
  envs=${cfg#* }
  [ "$envs" = "$cfg" ] && envs=""
  env $envs python bench.py --steps ${STEPS:-10} --warmup 2 --no-other-configs --no-cpu-baseline --no-verify --no-latency-probe --no-image-to-proof ${EXTRA} 2>/dev/null \
    | python -c "import json,sys; d=json.load(sys.stdin); s=(d.get('roofline') or {}).get('serialised') or {}; print('$label ms/proof %.4f device_ms/proof %.3f serialised %.4f' % (d['ms_per_proof'], d['device_ms_per_proof'], s.get('ms_per_proof', 0)))" >> $out
  tail -1 $out
done; done
python - <<PY
import collections
r = collections.defaultdict(list); s = collections.defaultdict(list)
for ln in open("$out"):
    f = ln.split()
    r[f[0]].append(float(f[2])); s[f[0]].append(float(f[6]))
for k in r:
    print("%-14s ms/proof mean %.4f min %.4f  (n=%d)   one-prover serialised mean %.4f" % (k, sum(r[k]) / len(r[k]), min(r[k]), len(r[k]), sum(s[k]) / len(s[k])))
PY
