#!/bin/bash
# MSM reduction of the latency configuration: lanes per EC addition x buckets per block.  ./tools/sweep_lanes.sh [runs]
for r in $(seq 1 ${1:-2}); do for cfg in "2 128" "4 64" "2 64" "4 128"; do
set -- $cfg
ZG_MSM_LANES=$1 ZG_MSM_RB=$2 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null > /tmp/ln.json; python -c "
import json; d=json.load(open('/tmp/ln.json')); k=d['single_proof_kernels_ms']; print('$cfg', 'lanes/bucket-block: single proof', round(d['create_proof_wall_s']*1e3,3), 'ms', {n:k[n] for n in ('msm_heavy','msm_bucket_scan','msm_bucket_sum','msm_finish')})"
done; done
