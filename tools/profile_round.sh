#!/bin/bash
# Round profile on the GPU box:  ./tools/profile_round.sh TAG     (outputs under gpurun_out/prof_TAG/)
#   1. rocprofv3 --kernel-trace --stats of the default bench command
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, no tracing flags
# Copy the summaries you want judged into profiles/ afterwards (tools/pmc_traffic.py builds the json).
set -e
TAG=${1:-cur}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-kernel-events --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-kernel-events --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.log
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
ls $OUT
