#!/bin/bash
# Which property of the nine-limb transform pass costs the throughput form what the pass gains alone (profiles/r05/ab_ntt9.txt, section 6):
# four builds through the headline at twelve provers, alternating, with the bench's power / clock figures.  The variant libraries
# (git-ignored, under tools/ab/): libzg_head.so = this tree; libzg_ntt9_w5.so = ntt9_pass_kernel with
# __attribute__((amdgpu_waves_per_eu(5))) (96 VGPRs, 20-36 B of spills); libzg_fat_LDS.so = ntt_pass_kernel launched with the nine-limb
# pass's 36 bytes of LDS per element (ELEM = 36 in launch_passes).
cd "$(dirname "$0")/.."
run() { # name lib ntt9
  ZG_HALO2_LIB=$PWD/tools/ab/$2 ZG_NTT9=$3 python3 bench.py --model tiny --provers 12 --steps 10 --warmup 3 --tail-only-headline --no-kernel-events 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); w=d.get('power',{})
print('$1: %.4f ms/proof  %.0f W %.0f MHz' % (d['ms_per_proof'], w.get('power_w_avg',0), w.get('sclk_mhz_avg',0)))"
}
for r in 1 2 3; do
  run "8-limb            " libzg_head.so 0
  run "9-limb 127 vgpr   " libzg_head.so 1
  run "9-limb  96 vgpr   " libzg_ntt9_w5.so 1
  run "8-limb, 36 B/elt LDS" libzg_fat_LDS.so 0
done
