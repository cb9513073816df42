#!/usr/bin/env python3
"""Copy the summaries of a `tools/profile_r02.sh TAG` run from gpurun_out/prof_TAG into profiles/r02/ and build
profiles/r02/pmc_traffic.json (HBM bytes per launch per kernel, VALU wave-instructions per proof):
    python tools/install_r02.py TAG
Counter passes: ONE prover, lock-step batches of 16 proofs, kernels serialised by the profiler.  FETCH_SIZE / WRITE_SIZE
are KB per dispatch; FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads -- MI355X_MICROARCH.md,
HBM section); bytes per launch = (2*FETCH + WRITE) * 1024, averaged over the launches of create_proof batches."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles", "r02")
os.makedirs(dst, exist_ok=True)
BATCH = 16


def one(pattern):
    """the newest match (gpurun merges into gpurun_out/ without clearing it: an older run's files may sit beside)"""
    return max(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)


def launch_name(kernel: str) -> str:
    """rocprof kernel name -> the label bench.py / ZG_LAUNCH uses"""
    k = re.sub(r"^(void )?zg::", "", kernel)
    m = re.match(r"ntt9?_pass_kernel<(\d+), (true|false), (true|false)>", k)
    if m:
        cols, first = m.group(2) == "true", m.group(3) == "true"
        return "ntt_cols" if cols else ("ntt_single" if first else "ntt_rows")
    k = k.split("(")[0].split("<")[0]
    k = re.sub(r"_kernel$", "", k)
    alias = {"gp_strip_scan": "grand_product_scan", "gp_strip_apply": "grand_product_apply", "gp_local": "grand_product_local",
             "gp_totals": "grand_product_totals", "gp_apply": "grand_product_apply", "kd_local": "kate_local",
             "kd_heads": "kate_heads", "kd_apply": "kate_apply",
             "dot": "eval_dot", "kd_strip": "kate_division",
             "pp_flags": "permute_flags", "pp_scan": "permute_scan", "pp_leftover": "permute_leftover",
             "pp_build": "permute_build", "random_and_blind": "random_poly", "evaluate_h9": "evaluate_h",
             "horner_combine_sets": "horner_combine", "msm_digits_naf": "msm_digits", "sort_global_fused": "sort_global", "gate_factor9": "gate_factor"}
    return alias.get(k, k)


def batches(path, counter):
    """the run's dispatches cut into create_proof batches at the kernel each batch starts with"""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out, cur = [], None
    for r in rows:
        n = launch_name(r["Kernel_Name"])
        if n == "random_poly":
            cur = []
            out.append(cur)
        if cur is not None:
            cur.append((n, float(r["Counter_Value"])))
    return [b for b in out if any(n == "kate_division" for n, _ in b)]  # complete ones


def per_launch(path, counter):
    s, c = defaultdict(float), defaultdict(int)
    for b in batches(path, counter):
        for n, v in b:
            s[n] += v
            c[n] += 1
    return {k: s[k] / c[k] for k in s}, c


fetch_p, write_p, valu_p = (one(f"pmc_{c}/*/*_counter_collection.csv") for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"))
fetch, nf = per_launch(fetch_p, "FETCH_SIZE")
write, _ = per_launch(write_p, "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU, three separate passes of `bench.py --steps 2 "
                "--warmup 1 --provers 1 --batch 16 --no-kernel-events --no-cpu-baseline --no-other-configs --no-verify "
                "--no-latency-probe` (tools/profile_r02.sh); FETCH/WRITE are KB per dispatch, FETCH_SIZE doubled per the gfx950 "
                "note in MI355X_MICROARCH.md; bytes per launch = (2*FETCH + WRITE)*1024 averaged over the launches inside "
                "create_proof batches; one launch serves 16 proofs",
       "proofs_per_launch": BATCH, "kernels": {}}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0))):
    out["kernels"][k] = {"launches": nf[k], "fetch_kb_raw": round(fetch[k], 1), "write_kb": round(write.get(k, 0.0), 1),
                         "hbm_bytes_per_launch": int((2 * fetch[k] + write.get(k, 0.0)) * 1024)}
vb = batches(valu_p, "SQ_INSTS_VALU")
per = defaultdict(float)
for b in vb:
    for n, v in b:
        per[n] += v / (len(vb) * BATCH)
total = sum(per.values())
out["valu"] = {"_note": "VALU wave-instructions one create_proof issues in the benchmarked form (lock-step batch of 16, split "
                        "extended domain): SQ_INSTS_VALU summed over the kernels of a batch / 16, averaged over the run's batches",
               "source": f"profiles/r02/{tag}_valu_instructions_per_proof.txt", "batches": len(vb),
               "wave_instructions_per_proof": total, "by_kernel": dict(sorted(per.items(), key=lambda kv: -kv[1]))}
json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
with open(os.path.join(dst, f"{tag}_valu_instructions_per_proof.txt"), "w") as f:
    f.write(f"SQ_INSTS_VALU per proof (wave-instructions), {len(vb)} batches of {BATCH}: total {total:.6g}\n")
    for n, v in sorted(per.items(), key=lambda kv: -kv[1]):
        f.write(f"{n:28s} {v:14.6g}  {100 * v / total:6.2f} %\n")
shutil.copy(one("trace/*/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_create_proof_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
shutil.copy(os.path.join(src, "bench_under_trace.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
for kind, path in (("fetch", fetch_p), ("write", write_p)):
    with open(os.path.join(dst, f"{tag}_pmc_{kind}_size.csv"), "w", newline="") as f:
        w = csv.writer(f)
        cols = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "Counter_Name", "Counter_Value"]
        w.writerow(cols)
        for r in csv.DictReader(open(path)):
            r["Kernel_Name"] = r["Kernel_Name"].split("(")[0]
            w.writerow([r[c] for c in cols])
b = json.load(open(os.path.join(dst, f"{tag}_bench.json")))
u = json.load(open(os.path.join(dst, f"{tag}_bench_under_rocprof.json")))
print(f"VALU per proof {total:.4g} over {len(vb)} batches; bench ms/proof {b['ms_per_proof']:.4f} (under rocprof {u['ms_per_proof']:.4f}); "
      f"dominant {b['roofline']['kernel']} avg launch {b['roofline']['avg_launch_ms']:.4f} ms (under rocprof {u['roofline']['avg_launch_ms']:.4f})")
for r in csv.DictReader(open(os.path.join(dst, f"{tag}_create_proof_kernel_stats.csv"))):
    if launch_name(r["Name"]) == b["roofline"]["kernel"]:
        print("rocprof:", r["Name"].split("(")[0], "avg us", float(r["AverageNs"]) / 1e3, "calls", r["Calls"])
