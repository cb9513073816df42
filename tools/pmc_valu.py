#!/usr/bin/env python3
"""Per-kernel VALU accounting from a `rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE`
run of bench.py (kernels are serialised under counter collection, so the sums are per-kernel totals):
    python tools/pmc_valu.py <counter_collection.csv> <proofs in the run>
Prints, per kernel, dispatches / proof, VALU instructions / proof and active-VALU cycles / proof."""
import csv
import re
import sys
from collections import defaultdict

path, proofs = sys.argv[1], float(sys.argv[2])
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in csv.DictReader(open(path)):
    name = re.sub(r"^(void )?zg::", "", r["Kernel_Name"]).split("(")[0].split("<")[0]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    acc[name]["ns"] += 0  # placeholder
    cnt[name].add(r["Dispatch_Id"])
    acc[name]["dur_" + r["Counter_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
counters = sorted({k for v in acc.values() for k in v if not k.startswith("dur_") and k != "ns"})
rows = []
for name, v in acc.items():
    dur = max((v[k] for k in v if k.startswith("dur_")), default=0.0)
    rows.append((name, len(cnt[name]) / proofs, dur / proofs / 1e6, [v.get(c, 0.0) / proofs for c in counters]))
rows.sort(key=lambda r: -r[3][counters.index("SQ_ACTIVE_INST_VALU")] if "SQ_ACTIVE_INST_VALU" in counters else -r[2])
print(f"{'kernel':28s} {'disp/proof':>10s} {'ms/proof':>9s} " + " ".join(f"{c:>22s}" for c in counters))
tot = [0.0] * len(counters)
for name, d, ms, vals in rows:
    print(f"{name[:28]:28s} {d:10.1f} {ms:9.3f} " + " ".join(f"{x:22.4g}" for x in vals))
    tot = [a + b for a, b in zip(tot, vals)]
print(f"{'TOTAL':28s} {'':10s} {'':9s} " + " ".join(f"{x:22.4g}" for x in tot))
