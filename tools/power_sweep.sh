#!/bin/bash
# Throughput, package power and shader clock against the number of provers sharing the chip:   ./tools/power_sweep.sh [MODEL] [PROVERS...]
# (the bench's own `power` object: hwmon sampled over the timed steps)
cd "$(dirname "$0")/.."
M=${1:-tiny}; shift
for p in ${@:-1 2 3 4 6 8 12 16}; do
  python3 bench.py --model $M --provers $p --steps $([ $p -le 2 ] && echo 30 || echo 12) --warmup 3 --tail-only-headline --no-kernel-events 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); w = d.get('power', {})
print('$M provers %2d: %.4f ms/proof  %6.1f W avg (max %6.1f of %d)  %6.1f MHz  %.3f J/proof' % ($p, d['ms_per_proof'], w.get('power_w_avg', 0), w.get('power_w_max', 0), w.get('power_cap_w', 0), w.get('sclk_mhz_avg', 0), w.get('joules_per_proof', 0)))"
done
