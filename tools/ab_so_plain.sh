#!/bin/bash
# Same-box A/B of the throughput bench between BUILDS of the library whose profiling structs may differ (no kernel events):
#   tools/ab_so_plain.sh A.so B.so [...]        REPS alternations (default 3), STEPS steps each
L=0g-halo2_amd/libzg_halo2.so
cp $L /tmp/zg_keep.so
out=gpurun_out/${OUT:-ab_so_plain}.txt
: > $out
for rep in $(seq 1 ${REPS:-3}); do for v in "$@"; do
  cp "$v" $L
  python bench.py --steps ${STEPS:-10} --warmup 2 --no-other-configs --no-cpu-baseline --no-verify --no-latency-probe --no-image-to-proof --no-kernel-events 2>/dev/null \
    | python -c "import json,sys; d=json.load(sys.stdin); print('$(basename $v) ms/proof %.4f' % d['ms_per_proof'])" >> $out
  tail -1 $out
done; done
cp /tmp/zg_keep.so $L
python - <<PY
import collections
r = collections.defaultdict(list)
for ln in open("$out"):
    f = ln.split(); r[f[0]].append(float(f[2]))
for k, v in r.items(): print("%-16s mean %.4f min %.4f (n=%d)" % (k, sum(v) / len(v), min(v), len(v)))
PY
