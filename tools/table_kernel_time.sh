#!/bin/bash
# Duration of msm_full_table_kernel (the digit-table build) under rocprofv3, for one or more builds of the library:
#   tools/table_kernel_time.sh A.so [B.so ...]
L=0g-halo2_amd/libzg_halo2.so
cp $L /tmp/zg_keep.so
R=$PWD
for v in "$@"; do
  cp "$v" $L
  rm -rf $R/gpurun_out/tk
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tk -- python3 $R/tools/table_build_time.py > $R/gpurun_out/tk.txt 2>&1)
  echo "$(basename $v): $(tail -2 $R/gpurun_out/tk.txt | tr '\n' ' ')"
  grep -h "msm_full_table" $R/gpurun_out/tk/*/*_kernel_stats.csv | cut -c1-200
done
cp /tmp/zg_keep.so $L
