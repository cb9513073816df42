#!/usr/bin/env python3
"""HBM bytes per launch of every kernel from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE):
    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
Counters are KB per dispatch; FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced
reads -- MI355X_MICROARCH.md, HBM section); bytes per launch = (2*FETCH + WRITE) * 1024, averaged."""
import csv
import json
import re
import sys
from collections import defaultdict


def launch_name(kernel: str) -> str:
    k = re.sub(r"^(void )?zg::", "", kernel)
    m = re.match(r"ntt_pass_kernel<(\d+), (true|false), (true|false)>", k)
    if m:
        cols, first = m.group(2) == "true", m.group(3) == "true"
        return "ntt_cols" if cols else ("ntt_single" if first else "ntt_rows")
    k = k.split("(")[0].split("<")[0]
    k = re.sub(r"_kernel$", "", k)
    alias = {"gp_local": "grand_product_local", "gp_totals": "grand_product_totals", "gp_apply": "grand_product_apply",
             "dot": "eval_dot", "kd_local": "kate_local", "kd_heads": "kate_heads", "kd_apply": "kate_apply",
             "pp_flags": "permute_flags", "pp_scan": "permute_scan", "pp_leftover": "permute_leftover",
             "pp_build": "permute_build", "random": "random_poly"}
    return alias.get(k, k)


def collect(path, counter):
    s, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = launch_name(r["Kernel_Name"])
        s[k] += float(r["Counter_Value"])
        n[k] += 1
    return s, n


fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
write, nw = collect(sys.argv[2], "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate passes of `bench.py --steps 2 --warmup 1 "
                "--streams 1 --no-kernel-events --no-cpu-baseline` (tools/profile_round.sh); counters are KB per "
                "dispatch; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md; bytes per launch = "
                "(2*FETCH + WRITE)*1024 averaged over the launches of the run",
       "kernels": {}}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0))):
    f, w = fetch[k] / max(nf[k], 1), write.get(k, 0.0) / max(nw.get(k, 1), 1)
    out["kernels"][k] = {"launches": nf[k], "fetch_kb_raw": round(f, 1), "write_kb": round(w, 1),
                         "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(out["kernels"]), "kernels")
