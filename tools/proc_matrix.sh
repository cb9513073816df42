#!/bin/bash
# processes x streams-per-process throughput matrix (diagnostic: is one process host-limited?)
#   ./tools/proc_matrix.sh "P S" "P S" ...     (env is passed through, e.g. ZG_SIDE_STREAM=0 ZG_MSM_C=14)
run() { # nproc streams
  local P=$1 S=$2; local pids=()
  for i in $(seq 1 $P); do python bench.py --steps ${STEPS:-12} --warmup 2 --streams $S --no-cpu-baseline --no-kernel-events 2>/dev/null > /tmp/m_$i.json & pids+=($!); done
  wait
  python - $P $S <<'PY'
import json,sys
P,S=int(sys.argv[1]),int(sys.argv[2]); tot=0
for i in range(1,P+1): tot+=json.load(open(f'/tmp/m_{i}.json'))["value"]
print(f"{P} proc x {S} streams: {3.6e6/tot:.3f} ms/proof  {tot:,.0f} proofs/h", flush=True)
PY
}
for ps in "$@"; do run $ps; done
