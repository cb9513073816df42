#!/usr/bin/env python3
"""Copy the summaries of a `tools/profile.sh TAG` run from gpurun_out/prof_TAG into profiles/ROUND/ and derive
    pmc_traffic.json     HBM bytes per launch per kernel (FETCH_SIZE / WRITE_SIZE passes, with the per-kernel FETCH_SIZE
                         correction that tools/fetch_calib.hip measured) and VALU wave-instructions per proof
    sq_issue.json/.txt   per kernel, alone on the chip: clock, waves per SIMD, the active / issue-stall / wait split of the
                         wave cycles, VALU issue utilisation -- the evidence behind "issue-saturated"
    serial_kernels.json  the serialised per-kernel table (one prover: kernels one at a time) of DESIGN.md section 4
    fetch_calibration.json
    affine_ab.json       (when the run had AFFINE=R) VALU instructions and HBM bytes per proof of the MSM kernels with and
                         without R rounds of batched-affine pre-reduction
    python tools/install_profile.py ROUND TAG [MODEL]        e.g. r05 v1 / r05 v1 medium
The tiny model's files carry no prefix (as in rounds 1-4); another MODEL's are MODEL_* (no shared-chip trace, calibration or
bench line for those: tools/profile.sh takes them for the tiny model only)."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

rnd, tag = sys.argv[1], sys.argv[2]
MODEL = sys.argv[3] if len(sys.argv) > 3 else "tiny"
PRE = "" if MODEL == "tiny" else MODEL + "_"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}" if MODEL == "tiny" else f"prof_{tag}_{MODEL}")
dst = os.path.join(root, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
# proofs per lock-step batch of the profiled runs (tools/profile.sh)
BATCH = int(os.environ.get("BATCH", {"tiny": "32", "large": "8"}.get(MODEL, "16")))
SIMDS = 256 * 4


def one(pattern):
    """the newest match (gpurun merges into gpurun_out/ without clearing it: an older run's files may sit beside)"""
    return max(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)


def launch_name(kernel: str) -> str:
    """rocprof kernel name -> the label bench.py / ZG_LAUNCH uses"""
    k = re.sub(r"^(void )?zg::", "", kernel)
    m = re.match(r"ntt_pass_kernel<(\d+), (true|false), (true|false)>", k)
    if m:
        cols, first = m.group(2) == "true", m.group(3) == "true"
        return "ntt_cols" if cols else ("ntt_single" if first else "ntt_rows")
    k = k.split("(")[0].split("<")[0]
    k = re.sub(r"_kernel$", "", k)
    alias = {"gp_strip_scan": "grand_product_scan", "gp_strip_apply": "grand_product_apply", "gp_local": "grand_product_local",
             "gp_totals": "grand_product_totals", "gp_apply": "grand_product_apply", "kd_local": "kate_local",
             "kd_heads": "kate_heads", "kd_apply": "kate_apply", "dot": "eval_dot", "kd_strip": "kate_division",
             "pp_flags": "permute_flags", "pp_scan": "permute_scan", "pp_scan_local": "permute_scan", "pp_scan_tiles": "permute_scan", "pp_leftover": "permute_leftover",
             "pp_build": "permute_build", "random_and_blind": "random_poly", "evaluate_h9": "evaluate_h",
             "horner_combine_sets": "horner_combine", "msm_digits_naf": "msm_digits", "sort_global_fused": "sort_global",
             "gate_factor9": "gate_factor", "combine9": "horner_combine", "dot9": "eval_dot", "msm_accumulate_full": "msm_accumulate_full",
             "aff_prefix": "msm_aff_prefix", "aff_apply": "msm_aff_apply", "aff_inv_up": "msm_aff_invert", "aff_inv_top": "msm_aff_invert",
             "aff_inv_down": "msm_aff_invert", "msm_heavy_groups": "msm_heavy", "witness_run_lds": "witness_run"}
    return alias.get(k, k)


def dispatches(path):
    """[(dispatch id, label, {counter: value}, duration ns)] in dispatch order"""
    by = {}
    for r in csv.DictReader(open(path)):
        d = by.setdefault(int(r["Dispatch_Id"]), [launch_name(r["Kernel_Name"]), {}, float(r["End_Timestamp"]) - float(r["Start_Timestamp"])])
        d[1][r["Counter_Name"]] = d[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [(i, *by[i]) for i in sorted(by)]


def batches(rows):
    """the run's dispatches cut into create_proof batches at the kernel each batch starts with; complete ones only"""
    out, cur = [], None
    # (image -> proof batches start with the witness program; from-resident ones with the random polynomial)
    start = "witness_run" if any(n == "witness_run" for _, n, _, _ in rows) else "random_poly"
    for _, n, c, ns in rows:
        if n == start:
            cur = []
            out.append(cur)
        if cur is not None:
            cur.append((n, c, ns))
    return [b for b in out if any(n == "kate_division" for n, _, _ in b)]


def per_launch(path):
    s, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(int)
    bs = batches(dispatches(path))
    for b in bs:
        for n, c, ns in b:
            cnt[n] += 1
            s[n]["ns"] += ns
            for k_, v in c.items():
                s[n][k_] += v
    return s, cnt, len(bs)


# ---- FETCH_SIZE calibration (tools/fetch_calib.hip): counter / true bytes for the library's three access shapes
calib = None
if os.path.exists(os.path.join(src, "calib_true.json")):
  true = json.load(open(os.path.join(src, "calib_true.json")))["true_bytes"]
  cal = defaultdict(list)
  for r in csv.DictReader(open(one("calib/*/*_counter_collection.csv"))):
    if r["Counter_Name"] == "FETCH_SIZE":
        cal[re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]].append(float(r["Counter_Value"]) * 1024)
  mean = lambda v: sum(v) / len(v)
  calib = {"stream16_16B_per_lane_coalesced": mean(cal["stream16"]) / true["stream16"],
         "gather_64B_record_per_lane": mean(cal["gather<4>"]) / true["gather<4>"],
         "gather_32B_record_per_lane": mean(cal["gather<2>"]) / true["gather<2>"]}
  json.dump({"_note": "rocprofv3 --pmc FETCH_SIZE over tools/fetch_calib.bin (2 GiB table, every byte read once): FETCH_SIZE * 1024 / "
                    "true bytes.  0.5 for 16-byte-per-lane streaming reads (the guide's halving: double the counter); 1.0 for "
                    "64-byte gathers (msm_accumulate's table points: the counter is exact, no doubling); 2.0 for 32-byte gathers "
                    "(every 32-byte record costs a 64-byte request).", "counter_over_true_bytes": calib},
          open(os.path.join(dst, "fetch_calibration.json"), "w"), indent=1)
# kernels whose reads are dominated by 64-byte gathers take the raw counter; everything else streams 16 B per lane
# (msm_accumulate_full gathers 64-byte points from the digit tables the same way -- ADVICE r3; the batched-affine kernels'
#  first round gathers the same table rows)
GATHER_KERNELS = {"msm_accumulate": 1.0, "msm_accumulate_full": 1.0}

fetch, nf, nbat = per_launch(one("pmc_FETCH_SIZE/*/*_counter_collection.csv"))
write, _, _ = per_launch(one("pmc_WRITE_SIZE/*/*_counter_collection.csv"))
out = {"_note": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of ONE prover making lock-step batches of {BATCH} "
                "(tools/profile.sh: kernels serialised); counters are KB per dispatch; bytes per launch = "
                "(fetch_correction * FETCH + WRITE) * 1024 averaged over the launches inside complete create_proof batches; "
                "fetch_correction = 2 for kernels that stream 16 B per lane (the gfx950 halving, MI355X_MICROARCH.md), 1 for "
                "msm_accumulate, whose reads are 64-byte gathers (profiles/{rnd}/fetch_calibration.json: the counter is exact there)",
       "proofs_per_launch": BATCH, "batches": nbat, "kernels": {}}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"])):
    f, w = fetch[k]["FETCH_SIZE"] / nf[k], write[k]["WRITE_SIZE"] / max(nf[k], 1)
    corr = 2.0 if k not in GATHER_KERNELS else GATHER_KERNELS[k]
    out["kernels"][k] = {"launches": nf[k], "fetch_kb_raw": round(f, 1), "write_kb": round(w, 1), "fetch_correction": corr,
                         "hbm_bytes_per_launch": int((corr * f + w) * 1024),
                         "hbm_bytes_per_launch_if_doubled": int((2 * f + w) * 1024)}

sq1, n1, nb1 = per_launch(one("pmc_sq1/*/*_counter_collection.csv"))
sq2, n2, _ = per_launch(one("pmc_sq2/*/*_counter_collection.csv"))
per = {k: sq1[k]["SQ_INSTS_VALU"] / (nb1 * BATCH) for k in sq1}
total = sum(per.values())
out["valu"] = {"_note": f"VALU wave-instructions one create_proof issues in the benchmarked form (lock-step batch of {BATCH}, split "
                        f"extended domain): SQ_INSTS_VALU summed over the kernels of a batch / {BATCH}, averaged over the run's batches",
               "source": f"profiles/{rnd}/{PRE}{tag}_valu_instructions_per_proof.txt", "batches": nb1,
               "wave_instructions_per_proof": total, "by_kernel": dict(sorted(per.items(), key=lambda kv: -kv[1]))}
json.dump(out, open(os.path.join(dst, PRE + "pmc_traffic.json"), "w"), indent=1)
with open(os.path.join(dst, f"{PRE}{tag}_valu_instructions_per_proof.txt"), "w") as f:
    f.write(f"SQ_INSTS_VALU per proof (wave-instructions), {nb1} batches of {BATCH}: total {total:.6g}\n")
    for n, v in sorted(per.items(), key=lambda kv: -kv[1]):
        f.write(f"{n:28s} {v:14.6g}  {100 * v / total:6.2f} %\n")

# ---- issue saturation, kernel by kernel (each alone on the chip)
issue = {"_note": "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU "
                  "SQ_INSTS_VALU GRBM_GUI_ACTIVE and a second pass with SQ_INSTS_VALU_INT64/INT32, LDS and VMEM counters; ONE prover, "
                  f"batches of {BATCH}, kernels serialised.  SQ_* cycle counters are quad-cycles (MI355X_MICROARCH.md); clock = "
                  "GRBM_GUI_ACTIVE / 8 XCDs / duration; waves_per_simd = 4 * SQ_WAVE_CYCLES / (1024 SIMDs * cycles); active / "
                  "issue_stall / wait = SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY over SQ_WAVE_CYCLES (they add up to 1); "
                  "valu_issue_util = 4 cycles * SQ_INSTS_VALU / (1024 SIMDs * cycles): the share of all SIMD issue cycles that a "
                  "VALU instruction occupies at 4 cycles per wave64 instruction (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU quad-cycles: "
                  "the counter's own price per instruction)", "kernels": {}}
rows = []
for k in sorted(sq1, key=lambda k: -sq1[k]["ns"]):
    c = sq1[k]
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    if cycles <= 0 or c["SQ_WAVE_CYCLES"] <= 0:
        continue
    rec = {"launches": n1[k], "avg_launch_ms": c["ns"] / n1[k] / 1e6, "us_per_proof": c["ns"] / (nb1 * BATCH) / 1e3,
           "clock_GHz": cycles / c["ns"],
           "waves_per_simd": 4 * c["SQ_WAVE_CYCLES"] / (SIMDS * cycles),
           "active": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stall": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
           "wait": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
           "valu_issue_util": 4 * c["SQ_INSTS_VALU"] / (SIMDS * cycles),
           "valu_instr_per_proof": c["SQ_INSTS_VALU"] / (nb1 * BATCH)}
    if k in sq2 and sq2[k].get("SQ_INSTS_VALU_INT64") is not None and c["SQ_INSTS_VALU"]:
        rec["int64_share_of_valu"] = sq2[k]["SQ_INSTS_VALU_INT64"] / max(1.0, c["SQ_INSTS_VALU"]) * n1[k] / max(1, n2[k])
        rec["lds_bank_conflict_share_of_lds_active"] = (sq2[k]["SQ_LDS_BANK_CONFLICT"] / sq2[k]["SQ_ACTIVE_INST_LDS"]
                                                        if sq2[k].get("SQ_ACTIVE_INST_LDS") else None)
    issue["kernels"][k] = rec
    rows.append((k, rec))
tot_ns = sum(sq1[k]["ns"] for k in sq1)
tot_cycles = sum(sq1[k]["GRBM_GUI_ACTIVE"] for k in sq1) / 8.0
tot_valu = sum(sq1[k]["SQ_INSTS_VALU"] for k in sq1)
issue["whole_proof_serialised"] = {"ms_per_proof": tot_ns / (nb1 * BATCH) / 1e6, "clock_GHz": tot_cycles / tot_ns,
                                   "valu_issue_util": 4 * tot_valu / (SIMDS * tot_cycles)}
json.dump(issue, open(os.path.join(dst, PRE + "sq_issue.json"), "w"), indent=1)
with open(os.path.join(dst, f"{PRE}{tag}_sq_issue.txt"), "w") as f:
    f.write(f"{'kernel':24s} {'us/proof':>9s} {'GHz':>5s} {'waves/SIMD':>10s} {'active':>7s} {'iss.stall':>9s} {'wait':>6s} {'VALU util':>9s} {'int64':>6s}\n")
    for k, r in rows:
        f.write(f"{k:24s} {r['us_per_proof']:9.1f} {r['clock_GHz']:5.2f} {r['waves_per_simd']:10.2f} {r['active']:7.2f} "
                f"{r['issue_stall']:9.2f} {r['wait']:6.2f} {r['valu_issue_util']:9.2f} {r.get('int64_share_of_valu', float('nan')):6.2f}\n")
    w = issue["whole_proof_serialised"]
    f.write(f"whole proof, kernels one at a time: {w['ms_per_proof']:.3f} ms/proof at {w['clock_GHz']:.2f} GHz, VALU issue utilisation {w['valu_issue_util']:.2f}\n")

# ---- serialised kernel table from the trace pass (no counters: the true durations)
ser = defaultdict(lambda: [0, 0.0])
trace_rows = []
for r in csv.DictReader(open(one("serial/*/*_kernel_trace.csv"))):
    trace_rows.append((int(r["Start_Timestamp"]), launch_name(r["Kernel_Name"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
trace_rows.sort()
tb = batches([(0, n, {}, ns) for _, n, ns in trace_rows])
for b in tb:
    for n, _, ns in b:
        ser[n][0] += 1
        ser[n][1] += ns
ser_total = sum(v[1] for v in ser.values())
serial = {"_note": f"rocprofv3 --kernel-trace of ONE prover making lock-step batches of {BATCH} (one stream: every kernel alone on the chip); "
                   "launches inside complete create_proof batches only", "batches": len(tb), "proofs_per_launch": BATCH,
          "device_us_per_proof": ser_total / (len(tb) * BATCH) / 1e3, "kernels": {}}
for k in sorted(ser, key=lambda k: -ser[k][1]):
    serial["kernels"][k] = {"launches_per_batch": ser[k][0] / len(tb), "avg_launch_us": ser[k][1] / ser[k][0] / 1e3,
                            "us_per_proof": ser[k][1] / (len(tb) * BATCH) / 1e3, "share": ser[k][1] / ser_total}
json.dump(serial, open(os.path.join(dst, PRE + "serial_kernels.json"), "w"), indent=1)

# ---- batched-affine A/B (tools/profile.sh with AFFINE=R): the MSM kernels' instructions and bytes per proof, both forms
if glob.glob(os.path.join(src, "aff_sq1/*/*_counter_collection.csv")):
    asq, an, anb = per_launch(one("aff_sq1/*/*_counter_collection.csv"))
    af, anf, _ = per_launch(one("aff_FETCH_SIZE/*/*_counter_collection.csv"))
    aw, _, _ = per_launch(one("aff_WRITE_SIZE/*/*_counter_collection.csv"))

    def msm_side(sq, nbat_sq, fe, wr, nfe, nbat_f):
        rows = {}
        for k in sq:
            if not k.startswith("msm_"):
                continue
            corr = 1.0 if (k in GATHER_KERNELS or k.startswith("msm_aff_a") or k.startswith("msm_aff_p")) else 2.0
            by = (corr * fe[k]["FETCH_SIZE"] + wr[k]["WRITE_SIZE"]) * 1024 / (nbat_f * BATCH) if k in fe else None
            rows[k] = {"valu_instr_per_proof": sq[k]["SQ_INSTS_VALU"] / (nbat_sq * BATCH), "us_per_proof_alone": sq[k]["ns"] / (nbat_sq * BATCH) / 1e3,
                       "hbm_bytes_per_proof": by}
        return {"kernels": dict(sorted(rows.items(), key=lambda kv: -kv[1]["valu_instr_per_proof"])),
                "msm_valu_instr_per_proof": sum(r["valu_instr_per_proof"] for r in rows.values()),
                "msm_us_per_proof_alone": sum(r["us_per_proof_alone"] for r in rows.values()),
                "msm_hbm_bytes_per_proof": sum(r["hbm_bytes_per_proof"] or 0 for r in rows.values()),
                "whole_proof_valu_instr": sum(sq[k]["SQ_INSTS_VALU"] for k in sq) / (nbat_sq * BATCH)}

    ab = {"_note": "ZG_MSM_AFFINE = 0 against R rounds of batched-affine pre-reduction (csrc/msm.hip, aff_* kernels): ONE prover, batches of "
                   f"{BATCH}, kernels serialised under counter collection; SQ_INSTS_VALU, FETCH_SIZE (x 2 for streaming kernels, x 1 for the "
                   "64-byte gathers of msm_accumulate and of the affine rounds), WRITE_SIZE per proof; us_per_proof_alone = the kernels' "
                   "durations under the counter pass",
          "rounds": int(os.environ.get("AFFINE", "2")),
          "without": msm_side(sq1, nb1, fetch, write, nf, nbat), "with": msm_side(asq, anb, af, aw, anf, anb)}
    json.dump(ab, open(os.path.join(dst, "affine_ab.json"), "w"), indent=1)
    print("affine A/B: MSM instructions per proof %.4g -> %.4g, MSM HBM bytes per proof %.4g -> %.4g, MSM us/proof alone %.1f -> %.1f" % (
        ab["without"]["msm_valu_instr_per_proof"], ab["with"]["msm_valu_instr_per_proof"], ab["without"]["msm_hbm_bytes_per_proof"],
        ab["with"]["msm_hbm_bytes_per_proof"], ab["without"]["msm_us_per_proof_alone"], ab["with"]["msm_us_per_proof_alone"]))

shutil.copy(one("serial/*/*_kernel_stats.csv"), os.path.join(dst, f"{PRE}{tag}_one_prover_kernel_stats.csv"))
tl = os.path.join(src, "lone_timeline.txt")
if os.path.exists(tl) and os.path.getsize(tl):
    kk = {"tiny": 14, "small": 15, "medium": 15, "large": 17}[MODEL]
    with open(os.path.join(dst, f"{PRE}lone_timeline_k{kk}.txt"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace of tools/lone_proof.py {MODEL} latency (one lone create_proof), cut by tools/timeline.py\n")
        f.write("# " + open(os.path.join(src, "lone.txt")).read().strip().replace("\n", "\n# ") + "\n")
        f.write(open(tl).read())
print(open(os.path.join(dst, f"{PRE}{tag}_sq_issue.txt")).read())
print(f"VALU per proof {total:.4g} over {nb1} batches of {BATCH}")
top = sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:6]
for k, v in top:
    print(f"  HBM bytes per launch {k:22s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB")
if MODEL == "tiny":
    shutil.copy(one("trace/*/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_create_proof_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench_line.json"))
    shutil.copy(os.path.join(src, "bench_detail.json"), os.path.join(dst, f"{tag}_bench_detail.json"))
    shutil.copy(os.path.join(src, "bench_under_trace.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
    b = json.load(open(os.path.join(dst, f"{tag}_bench_line.json")))
    u = json.load(open(os.path.join(dst, f"{tag}_bench_under_rocprof.json")))
    print(f"calibration {calib}")
    print(f"bench ms/proof {b['ms_per_proof']:.4f} (under rocprof {u['ms_per_proof']:.4f}); "
          f"dominant {b['roofline']['kernel']} avg launch {b['roofline']['avg_launch_ms']:.4f} ms (under rocprof {u['roofline']['avg_launch_ms']:.4f})")
    for r in csv.DictReader(open(os.path.join(dst, f"{tag}_create_proof_kernel_stats.csv"))):
        if launch_name(r["Name"]) == b["roofline"]["kernel"]:
            print("rocprof (shared chip):", r["Name"].split("(")[0], "avg us", float(r["AverageNs"]) / 1e3, "calls", r["Calls"])
