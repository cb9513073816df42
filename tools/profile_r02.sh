#!/bin/bash
# Round-2 profile on the GPU box:  ./tools/profile_r02.sh TAG     (outputs under gpurun_out/prof_TAG/)
#   1. rocprofv3 --kernel-trace --stats of the default bench command (12 provers x lock-step batches of 16)
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, three separate passes with no tracing flags, of ONE prover
#      making batches of 16 (kernels are serialised under counter collection; a launch = 16 proofs as in the timed run)
#   3. the bench line itself, without the profiler
# tools/install_r02.py TAG copies the summaries into profiles/r02/ and builds pmc_traffic.json.
set -e
TAG=${1:-cur}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
PMC_ARGS="--steps 2 --warmup 1 --provers 1 --batch 16 --no-kernel-events --no-cpu-baseline --no-other-configs --no-verify --no-latency-probe --no-image-to-proof"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-other-configs > $OUT/bench_under_trace.json 2> $OUT/trace.log
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py $PMC_ARGS > /dev/null 2> $OUT/pmc_$c.log
done
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
ls $OUT
