#!/bin/bash
# Round 4's library (tools/ab/libzg_r4final.so: the .so of commit 79186a9) against this tree's, alternating processes on one box:
# the bench's headline step image -> proof and the same provers from resident columns.   ./tools/ab_r4_vs_r5.sh [ROUNDS]
cd "$(dirname "$0")/.."
N=${1:-3}
for r in $(seq $N); do
  for lib in tools/ab/libzg_r4final.so 0g-halo2_amd/libzg_halo2.so; do
    for mode in "" "--from-resident"; do
      ZG_HALO2_LIB=$PWD/$lib python3 bench.py --steps 10 --warmup 3 --tail-only-headline --no-kernel-events $mode 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$(basename $lib) ${mode:-image->proof} round $r: %.4f ms/proof' % d['ms_per_proof'])"
    done
  done
done
