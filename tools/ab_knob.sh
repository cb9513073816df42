#!/bin/bash
# Same-box A/B of one knob through the bench's headline (image -> proof), alternating processes:
#   ./tools/ab_knob.sh KNOB V0 V1 [MODEL] [ROUNDS]      -> ms/proof with 12 provers and with ONE prover, per value and round
cd "$(dirname "$0")/.."
K=$1; A=$2; B=$3; M=${4:-tiny}; N=${5:-3}
for r in $(seq $N); do
  for v in $A $B; do
    for p in 12 1; do
      env $K=$v python3 bench.py --model $M --provers $p --steps $([ $p = 1 ] && echo 12 || echo 10) --warmup 3 --tail-only-headline --no-kernel-events 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$M $K=$v provers $p round $r: %.4f ms/proof' % d['ms_per_proof'])"
    done
  done
done
