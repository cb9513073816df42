#!/bin/bash
# Power, clock and throughput of the headline per value of one knob:   ./tools/power_ab.sh KNOB V0 V1 [MODEL]
# (rocm-smi sampled every 0.5 s while the bench's timed steps run: is the chip at its power cap, and at which clock?)
cd "$(dirname "$0")/.."
K=$1; A=$2; B=$3; M=${4:-tiny}
for v in $A $B $A $B; do
  ( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.5; done ) > /tmp/smi_$v.txt &
  S=$!
  env $K=$v python3 bench.py --model $M --provers 12 --steps 40 --warmup 5 --tail-only-headline --no-kernel-events 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$M $K=$v: %.4f ms/proof' % d['ms_per_proof'])"
  kill $S; wait $S 2>/dev/null
  python3 - $v <<'PY'
import json, sys
rows = []
for ln in open(f"/tmp/smi_{sys.argv[1]}.txt"):
    try:
        d = json.loads(ln)
    except Exception:
        continue
    c = d.get("card0", {})
    p = [float(v) for k, v in c.items() if "ower" in k and str(v).replace(".", "", 1).isdigit()]
    s = [v for k, v in c.items() if "sclk" in k.lower()]
    rows.append((p[0] if p else None, s[0] if s else None))
busy = [r for r in rows if r[0] and r[0] > 500]
print("  samples", len(rows), "busy", len(busy), "power W avg/max", round(sum(r[0] for r in busy) / max(len(busy), 1)), max([r[0] for r in busy] or [0]),
      "sclk", sorted(set(r[1] for r in busy))[:6])
PY
done
