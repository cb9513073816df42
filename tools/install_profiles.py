#!/usr/bin/env python3
"""Copy the summaries of a `tools/profile_round.sh TAG` run (+ the SQ_INSTS_VALU pass) from gpurun_out/prof_TAG
into profiles/r01/ and refresh profiles/r01/pmc_traffic.json:  python tools/install_profiles.py TAG [OLD_TAG]"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict

tag = sys.argv[1]
old = sys.argv[2] if len(sys.argv) > 2 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles", "r01")


def one(pattern):
    return glob.glob(os.path.join(src, pattern))[0]


fetch, write, valu = one("pmc_fetch/runc/*_counter_collection.csv"), one("pmc_write/runc/*_counter_collection.csv"), \
    one("pmc_valu/runc/*_counter_collection.csv")
subprocess.check_call([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), fetch, write,
                       os.path.join(dst, "pmc_traffic.json")])
shutil.copy(one("trace/runc/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_create_proof_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
shutil.copy(os.path.join(src, "bench_under_trace.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
with open(os.path.join(dst, f"{tag}_valu_instructions_per_proof.txt"), "w") as f:
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "pmc_valu.py"), valu, "12"], stdout=f)
for kind, path in (("fetch", fetch), ("write", write)):
    with open(os.path.join(dst, f"{tag}_pmc_{kind}_size.csv"), "w", newline="") as f:
        w = csv.writer(f)
        cols = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "Counter_Name", "Counter_Value",
                "Start_Timestamp", "End_Timestamp"]
        w.writerow(cols)
        for r in csv.DictReader(open(path)):
            r["Kernel_Name"] = r["Kernel_Name"].split("(")[0]
            w.writerow([r[c] for c in cols])
if old:
    for f in glob.glob(os.path.join(dst, f"{old}_pmc_*")) + [os.path.join(dst, f"{old}_valu_instructions_per_proof.txt")]:
        if os.path.exists(f):
            os.remove(f)

# VALU wave-instructions per proof: the single-stream run is cut into proofs at the kernel every proof starts with
# (random_and_blind); a proof is a throughput-configuration one when its MSM reduction ran in the one-lane form
rows = [r for r in csv.DictReader(open(valu)) if r["Counter_Name"] == "SQ_INSTS_VALU"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
proofs, cur = [], None
for r in rows:
    n = re.sub(r"^(void )?zg::", "", r["Kernel_Name"]).split("(")[0]
    if n.startswith("random_"):
        cur = defaultdict(float)
        proofs.append(cur)
    if cur is not None:
        cur[n] += float(r["Counter_Value"])
proofs = [q for q in proofs if any(k.startswith("msm_bucket_sum_kernel") for k in q)]  # (complete ones)
thr_proofs = [q for q in proofs if any(re.match(r"msm_bucket_sum_kernel<1,", k) for k in q)]
lat_proofs = [q for q in proofs if q not in thr_proofs]
nt, nl = len(thr_proofs), len(lat_proofs)
per = defaultdict(float)
for q in thr_proofs:
    for k, v in q.items():
        per[k] += v / nt
thr = sum(per.values())
lat = sum(sum(q.values()) for q in lat_proofs) / max(nl, 1)
p = os.path.join(dst, "pmc_traffic.json")
d = json.load(open(p))
d["valu"] = {
    "_note": f"rocprofv3 --pmc SQ_INSTS_VALU on a single-stream run (profiles/r01/{tag}_valu_instructions_per_proof.txt lists the "
             "run's per-kernel totals; the MSM kernels are attributed by their template instantiation): VALU "
             "wave-instructions one create_proof issues in the throughput configuration, set-up kernels excluded",
    "wave_instructions_per_proof": thr, "latency_configuration": lat,
    "top": dict(sorted(per.items(), key=lambda kv: -kv[1])[:7])}
json.dump(d, open(p, "w"), indent=1)
print(f"proofs in the VALU run: {nt} throughput, {nl} latency; VALU per proof {thr:.4g} / {lat:.4g}")
b = json.load(open(os.path.join(dst, f"{tag}_bench.json")))
u = json.load(open(os.path.join(dst, f"{tag}_bench_under_rocprof.json")))
print("bench: ms/proof", b["ms_per_proof"], "single proof s", b["create_proof_wall_s"], "accumulate ms", b["roofline"]["avg_launch_ms"],
      "| under rocprof:", u["ms_per_proof"], u["roofline"]["avg_launch_ms"])
for r in csv.DictReader(open(os.path.join(dst, f"{tag}_create_proof_kernel_stats.csv"))):
    if "accumulate" in r["Name"]:
        print("rocprof:", r["Name"].split("(")[0], "avg us", float(r["AverageNs"]) / 1e3, "calls", r["Calls"])
