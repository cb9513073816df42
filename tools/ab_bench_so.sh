#!/bin/bash
# A/B of the throughput bench between builds of the library: ./tools/ab_bench_so.sh A.so B.so [...]   (three rounds, alternating)
L=0g-halo2_amd/libzg_halo2.so
cp $L /tmp/zg_keep.so
out=gpurun_out/ab_bench_so.txt
: > $out
for rep in 1 2 3; do for v in "$@"; do
  cp "$v" $L
  python bench.py --steps ${STEPS:-20} --warmup 2 --no-other-configs --no-cpu-baseline --no-verify --no-latency-probe --no-image-to-proof 2>/dev/null \
    | python -c "import json,sys; d=json.load(sys.stdin); s=d['roofline'].get('serialised') or {}; print('$(basename $v) ms/proof %.4f device_ms/proof %.3f serialised %.4f' % (d['ms_per_proof'], d['device_ms_per_proof'], s.get('ms_per_proof', 0)))" >> $out
done; done
cp /tmp/zg_keep.so $L
cat $out
