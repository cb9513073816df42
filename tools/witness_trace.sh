#!/bin/bash
# Where a level of the witness program's replay spends its cycles (GPU box): builds csrc/witness.hip with -DZG_WITNESS_TRACE (lane 0
# of image 0 stamps s_memtime at: 1 level head, 2 its operation done, 3 barrier passed, 4 epoch's levels done, 5 deferred store
# issued, 6 next prefetch issued), links a second library beside the product's and prints the per-segment cycle sums.
#   ./tools/witness_trace.sh [model]
set -e
cd "$(dirname "$0")/../0g-halo2_amd"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../include -DZG_WITNESS_TRACE -c csrc/witness.hip -o /tmp/witness_trace.o
mkdir -p ../tools/ab
hipcc -shared -fPIC --offload-arch=gfx950 -o ../tools/ab/libzg_trace.so csrc/ctx.o csrc/msm.o csrc/ntt.o csrc/params.o csrc/poly.o csrc/prover.o csrc/sort.o /tmp/witness_trace.o
cd ..
ZG_HALO2_LIB=$PWD/tools/ab/libzg_trace.so python3 tools/witness_time.py ${1:-tiny} 2> /tmp/witness_trace.txt > /dev/null
python3 - <<EOF
import collections
rows = [l.split() for l in open("/tmp/witness_trace.txt") if l.startswith("T ")]
agg = collections.defaultdict(list)
for _, tag, dt in rows:
    agg[int(tag)].append(int(dt))
names = {1: "level head (level table read, loop)", 2: "the lane's operation (or none)", 3: "barrier passed", 4: "epoch's levels done",
         5: "epoch wait + deferred store issued", 6: "prefetch issued"}
print("cycles between trace points, lane 0 of image 0, one witness_run (each trace point itself costs ~170 cycles)")
for t, v in sorted(agg.items()):
    print(f"  {t} {names[t]:40s} n {len(v):4d}  mean {sum(v) / len(v):8.1f}  max {max(v):7d}  sum {sum(v):8d}")
print("  total", sum(sum(v) for v in agg.values()))
EOF
