#!/bin/bash
# provers x batch for the k = 15 / k = 17 configurations (image -> proof, headline only):  ./tools/sweep_models.sh   (GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5
for cfg in "small 12 16" "small 12 32" "small 16 16" "medium 12 16" "medium 12 32" "medium 16 16" "medium 16 24" "large 12 8" "large 16 8"; do
  set -- $cfg
  python3 bench.py --model $1 --provers $2 --batch $3 --steps 6 --warmup 2 --tail-only-headline --no-kernel-events 2> gpurun_out/r5/sweep.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1 provers $2 batch $3: %.4f ms/proof' % d['ms_per_proof'])"
done
