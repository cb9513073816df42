#!/usr/bin/env python3
"""Timeline of ONE proof from a rocprofv3 --kernel-trace csv of a single-stream run:
    python tools/timeline.py <kernel_trace.csv> [proof index from the end, default 1]
Prints every kernel of that proof with start offset, duration and the idle gap before it."""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"^(void )?zg::", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
# a proof starts at random_kernel (vanishing argument's random polynomial is generated first)
starts = [i for i, r in enumerate(rows) if r[2].startswith("random_")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
i0 = starts[-which - 1]
i1 = starts[-which]
t0 = rows[i0][0]
prev_end = t0
busy = gap = 0
for s, e, n in rows[i0:i1]:
    g = max(0, s - prev_end)
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {g / 1e3:7.1f}  {n}")
    busy += e - s
    gap += g
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, idle gaps {gap / 1e3:.1f} us, {i1 - i0} launches")
