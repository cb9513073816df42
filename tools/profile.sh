#!/bin/bash
# The round's profile set on the GPU box:  [MODEL=tiny|small|medium|large] [BATCH=n] ./tools/profile.sh TAG
# (outputs under gpurun_out/prof_TAG[_MODEL]/; tools/install_profile.py ROUND TAG [MODEL] files the summaries under profiles/ROUND/,
#  the tiny model's without a prefix, the others' as MODEL_*).  Every pass profiles the HEADLINE form of bench.py: image -> proof
#  (zg_prover_prove_images: witness program + create_proof), lock-step batches of $BATCH (default 32 / 16 / 16 / 8 by model).
#   1. SERIALISED kernel stats: rocprofv3 --kernel-trace --stats of ONE prover (one stream: every kernel alone on the chip) -- the
#      per-kernel table of DESIGN.md section 4
#   2. (tiny only) the same trace of the default bench command (12 provers sharing the chip): the roofline object's cross-check
#   3. counter passes of ONE prover (kernels serialised under counter collection), separate passes, no tracing flags:
#        sq1: SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
#             SQ_INSTS_VALU + GRBM_GUI_ACTIVE      (issue saturation: active / wait split of the wave cycles, clock)
#        sq2: SQ_INST_CYCLES_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS
#             SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT
#        FETCH_SIZE, WRITE_SIZE
#   4. (tiny only) tools/fetch_calib.bin under FETCH_SIZE: what the counter reports for 16-B streaming reads and 64-B / 32-B gathers
#   5. (tiny only) the bench line itself, without the profiler
#   6. (AFFINE=R, optional) the sq1 / FETCH_SIZE / WRITE_SIZE passes again with ZG_MSM_AFFINE=R
#   7. rocprofv3 --kernel-trace of tools/lone_proof.py MODEL latency: the timeline of one lone proof (tools/timeline.py)
# ZG_LAT_GATE must be OFF under counter collection (rocprofv3 --pmc serialises kernels across queues: a spinning gate kernel would
# hold back the side-stream work the host waits for); the passes below run no lone probe, and bench.py's probe checks for itself.
set -e
[ -x tools/fetch_calib.bin ] || hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/fetch_calib.bin
TAG=${1:-cur}
MODEL=${MODEL:-tiny}
case $MODEL in tiny) DB=32;; large) DB=8;; *) DB=16;; esac
BATCH=${BATCH:-$DB}
R=$PWD
if [ "$MODEL" = tiny ]; then OUT=$R/gpurun_out/prof_$TAG; else OUT=$R/gpurun_out/prof_${TAG}_$MODEL; fi
mkdir -p $OUT
ONE="--model $MODEL --steps 3 --warmup 1 --provers 1 --batch $BATCH --no-kernel-events --tail-only-headline"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 $R/bench.py $ONE > $OUT/serial_bench.json 2> $OUT/serial.log
echo "serial trace done"
if [ "$MODEL" = tiny ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-latency-probe --no-from-resident > $OUT/bench_under_trace.json 2> $OUT/trace.log
  echo "shared trace done"
fi
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq1 -- python3 $R/bench.py $ONE > /dev/null 2> $OUT/pmc_sq1.log
echo "sq1 done"
rocprofv3 --pmc SQ_INST_CYCLES_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py $ONE > /dev/null 2> $OUT/pmc_sq2.log
echo "sq2 done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py $ONE > /dev/null 2> $OUT/pmc_$c.log
  echo "$c done"
done
if [ "$MODEL" = tiny ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- $R/tools/fetch_calib.bin > $OUT/calib_true.json 2> $OUT/calib.log
  echo "calibration done"
fi
if [ -n "$AFFINE" ]; then
  export ZG_MSM_AFFINE=$AFFINE
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/aff_sq1 -- python3 $R/bench.py $ONE > /dev/null 2> $OUT/aff_sq1.log
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/aff_$c -- python3 $R/bench.py $ONE > /dev/null 2> $OUT/aff_$c.log
  done
  unset ZG_MSM_AFFINE
  echo "affine passes done"
fi
# 7. the kernel timeline of a LONE proof (latency form; digit tables where the size has them): tools/timeline.py cuts the last proof
#    out of the trace (ZG_LAT_GATE=0: under the profiler a launch costs the host ~20 us, and the gated schedule launches the next
#    phase BEFORE it reads this one's results -- the trace would show the profiler's overhead, not the schedule)
export ZG_LAT_GATE=0
rocprofv3 --kernel-trace --output-format csv -d $OUT/lone -- python3 $R/tools/lone_proof.py $MODEL latency > $OUT/lone.txt 2> $OUT/lone.log
unset ZG_LAT_GATE
echo "lone trace done"
cd $R
python3 tools/timeline.py $(ls -t $OUT/lone/*/*_kernel_trace.csv | head -1) > $OUT/lone_timeline.txt 2>> $OUT/lone.log || true
if [ "$MODEL" = tiny ]; then
  python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
  cp bench_detail.json $OUT/bench_detail.json
fi
# the raw counter / trace dumps are large (tens of MB) and gpurun merges at most 64 MiB back: keep what install_profile.py reads
find $OUT -name '*_agent_info.csv' -delete
ls $OUT
