#!/bin/bash
# Two-rank rehearsal of `bench.py --mode shard-msm` on ONE GPU (this pool has one per box): both ranks on cuda:0, gloo
# instead of RCCL (which refuses two ranks on one device), 4 provers per rank, each with an exchange group of its own.
#   tools/rehearse_shard.sh TAG [model ...]     -> gpurun_out/shard_TAG_<model>.json
set -e
TAG=${1:-cur}; shift || true
MODELS=${@:-tiny medium}
export ZG_BENCH_DEVICE=0 ZG_BENCH_BACKEND=gloo
P=29710
for m in $MODELS; do
  P=$((P+1))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $P bench.py \
    --gpus 2 --mode shard-msm --provers 4 --model $m --steps 5 --warmup 2 --no-cpu-baseline \
    > gpurun_out/shard_${TAG}_$m.json 2> gpurun_out/shard_${TAG}_$m.err
  python - <<PY
import json
d=json.load(open("gpurun_out/shard_${TAG}_$m.json"))
print("$m", "ms/proof", round(d["ms_per_proof"],4), "provers", d["provers_per_gpu"], "batch", d["batch"], "exchange", d["exchange"], "rccl_ranks", d["rccl_ranks"], "verified", d.get("verified"))
PY
done
