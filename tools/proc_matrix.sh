#!/bin/bash
# processes x streams-per-process throughput matrix (diagnostic: is one process host-limited?)
run() { # nproc streams
  local P=$1 S=$2; local pids=()
  for i in $(seq 1 $P); do python bench.py --steps 96 --warmup 16 --streams $S --no-cpu-baseline --no-kernel-events 2>/dev/null > /tmp/m_$i.json & pids+=($!); done
  wait
  python - $P $S <<'PY'
import json,sys
P,S=int(sys.argv[1]),int(sys.argv[2]); tot=0
for i in range(1,P+1): tot+=json.load(open(f'/tmp/m_{i}.json'))["value"]
print(f"{P} proc x {S} streams: {3.6e6/tot:.3f} ms/proof  {tot:,.0f} proofs/h")
PY
}
run 1 1; run 1 2; run 2 1; run 1 4; run 2 2; run 4 1; run 1 8; run 2 4; run 4 2
