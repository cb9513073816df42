#!/usr/bin/env python3
"""Prints the tables of DESIGN.md sections 4 - 6 from the committed profile files of a round, so that the document's
numbers are regenerated, not retyped:   python tools/design_tables.py r04

  kernels     per kernel ALONE on the chip: us per proof, share, launches per batch x average, VALU issue utilisation, the
              active / issue-stall / wait split (serial_kernels.json: trace pass; sq_issue.json: counter pass), algorithmic
              GB/s from the bench line's serialised pass
  families    SURVEY 8d units vs counter bytes (final_bench.json: roofline.families)
  shard       the projected compute leg of one rank (shard_compute_leg.json)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = os.path.join(ROOT, "profiles", rnd)


def load(name):
    try:
        return json.load(open(os.path.join(P, name)))
    except OSError:
        return None


ser, sq, line = load("serial_kernels.json"), load("sq_issue.json"), load("final_bench.json")
if ser and sq:
    alone = ((line or {}).get("roofline") or {}).get("serialised", {}).get("kernels", {})
    print(f"### kernels alone on the chip ({rnd}: serial_kernels.json, sq_issue.json; GB/s from final_bench.json roofline.serialised)\n")
    print("| kernel | µs per proof | share | launches per batch × avg µs | VALU issue util. | active / issue-stall / wait | waves / SIMD | algorithmic GB/s (of 8 000) |")
    print("|---|---|---|---|---|---|---|---|")
    other = 0.0
    for k, v in ser["kernels"].items():
        if v["share"] < 0.004:
            other += v["us_per_proof"]
            continue
        q = sq["kernels"].get(k, {})
        a = alone.get(k, {})
        split = f"{q['active']:.2f} / {q['issue_stall']:.2f} / {q['wait']:.2f}" if "active" in q else ""
        print(f"| `{k}` | {v['us_per_proof']:.1f} | {100 * v['share']:.1f} % | {v['launches_per_batch']:.0f} × {v['avg_launch_us']:.0f} | "
              f"{q.get('valu_issue_util', float('nan')):.2f} | {split} | {q.get('waves_per_simd', float('nan')):.2f} | "
              + (f"{a['algo_GBps']:.0f} ({a['frac_of_hbm_peak']:.3f}) |" if "algo_GBps" in a else "— |"))
    w = sq["whole_proof_serialised"]
    print(f"| everything else | {other:.1f} | | | | | | |")
    print(f"| **whole proof, kernels one at a time** | **{ser['device_us_per_proof']:.0f}** | | | {w['valu_issue_util']:.2f} | | | |\n")

if line:
    r = line["roofline"]
    proofs = line["steps"] * line["proofs_per_step"]
    print(f"### kernel families, timed region of the committed line ({rnd}/final_bench.json; {line['ms_per_proof']:.4f} ms/proof)\n")
    print("| family | share of device time | algorithmic bytes per proof | basis | counter bytes per proof | counter ÷ algorithmic | algorithmic GB/s |")
    print("|---|---|---|---|---|---|---|")
    for k, f in r["families"].items():
        cb = f"{f['counter_bytes'] / proofs / 1e6:.0f} MB" if f.get("counter_bytes") else "—"
        ratio = f"{f['counter_over_algorithmic_bytes']:.1f}×" if f.get("counter_over_algorithmic_bytes") else "—"
        print(f"| {k} | {100 * f['share_of_device_time']:.1f} % | {f['algorithmic_bytes'] / proofs / 1e6:.1f} MB | {f['basis']} | {cb} | {ratio} | "
              f"{f['algo_GBps']:.0f} |")
    print(f"\nSURVEY 8d per proof: {line['algorithmic_bytes_per_proof'] / 1e6:.1f} MB by the formula, "
          f"{line['algorithmic_bytes_per_proof_charged'] / 1e6:.1f} MB charged by the library launch by launch.\n")

sh = load("shard_compute_leg.json")
if sh:
    print(f"### one rank's compute leg of a point-range-sharded step ({rnd}/shard_compute_leg.json; no collective; not a scaling curve)\n")
    print("| model | k | world | points per rank | ms / proof | × vs world 1 | device ms / proof: msm | ntt | evaluate_h | the rest | msm_accumulate | strip + strip_sum + heavy + scan |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in sh["rows"]:
        f = r["family_device_ms_per_proof"]
        m = r["msm_kernels_device_ms_per_proof"]
        rest = sum(v for k, v in f.items() if k not in ("msm", "ntt", "evaluate_h"))
        red = sum(m.get(k, 0.0) for k in ("msm_strip", "msm_strip_sum", "msm_heavy", "msm_scan"))
        print(f"| {r['model'].replace('model_', '')} | {r['k']} | {r['world']} | {r['points_per_rank']} | {r['ms_per_proof']:.3f} | "
              f"{r['speedup_vs_world_1']:.2f} | {f.get('msm', 0):.2f} | {f.get('ntt', 0):.2f} | {f.get('evaluate_h', 0):.2f} | {rest:.2f} | "
              f"{m.get('msm_accumulate', 0):.2f} | {red:.2f} |")
