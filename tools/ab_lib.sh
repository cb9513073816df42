#!/bin/bash
# Two builds of the library through the bench's headline, alternating processes on one box:
#   ./tools/ab_lib.sh LIB_A LIB_B [MODEL] [ROUNDS]     -> ms/proof with 12 provers and with ONE prover
cd "$(dirname "$0")/.."
A=$1; B=$2; M=${3:-tiny}; N=${4:-3}
for r in $(seq $N); do
  for lib in $A $B; do
    for p in 12 1; do
      ZG_HALO2_LIB=$PWD/$lib python3 bench.py --model $M --provers $p --steps $([ $p = 1 ] && echo 12 || echo 10) --warmup 3 --tail-only-headline --no-kernel-events 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$M $(basename $lib) provers $p round $r: %.4f ms/proof' % d['ms_per_proof'])"
    done
  done
done
