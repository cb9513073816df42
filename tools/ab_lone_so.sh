#!/bin/bash
# A/B of a lone proof between builds of the library: ./tools/ab_lone_so.sh MODEL A.so B.so [...]   (two rounds each, alternating)
model=$1; shift
L=0g-halo2_amd/libzg_halo2.so
cp $L /tmp/zg_keep.so
out=gpurun_out/ab_lone_so.txt
: > $out
for rep in 1 2 3; do for v in "$@"; do
  cp "$v" $L
  echo "$(basename $v) $(python tools/lone_proof.py $model latency 2>/dev/null | cut -c1-120)" >> $out
done; done
cp /tmp/zg_keep.so $L
cat $out
