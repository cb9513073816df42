#!/bin/bash
cd "$(dirname "$0")/.."
for r in 1 2; do for w in 14 15 16; do
  ZG_MSM_NAF=$w python3 bench.py --model tiny --provers 12 --steps 10 --warmup 3 --tail-only-headline --no-kernel-events 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); w=d.get('power',{})
print('ZG_MSM_NAF=$w round $r: %.4f ms/proof  %.0f W %.0f MHz' % (d['ms_per_proof'], w.get('power_w_avg',0), w.get('sclk_mhz_avg',0)))"
done; done
