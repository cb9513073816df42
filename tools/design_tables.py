#!/usr/bin/env python3
"""Prints the tables of DESIGN.md sections 4 - 7 from the committed profile files of a round, so that the document's
numbers are regenerated, not retyped:   python tools/design_tables.py r05 [section ...]

  kernels     per kernel ALONE on the chip, one table per model that has a profile set (tiny: no prefix; medium_ / large_):
              us per proof, share, launches per batch x average, VALU issue utilisation, the active / issue-stall / wait split
              (<model>_serial_kernels.json: trace pass; <model>_sq_issue.json: counter pass), counter HBM bytes per launch against
              the kernel's algorithmic bytes (<model>_pmc_traffic.json; final_bench_detail.json)
  families    SURVEY 8d units vs counter bytes (final_bench_detail.json: roofline.families)
  models      the other configurations' lines (final_bench_detail.json: other_configs)
  lone        the lone-proof pairs (final_bench_detail.json: lone)
  shard       the projected compute leg of one rank, whole proofs (shard_compute_leg.json; round 4's file when this round has none)
  msm         ... of --mode msm-only (msm_only_compute_leg.json)
  witness     witness_run in its two forms (witness_run_forms.json)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
want = set(sys.argv[2:])
P = os.path.join(ROOT, "profiles", rnd)


def load(name, fallback_rounds=()):
    for d in (P,) + tuple(os.path.join(ROOT, "profiles", r) for r in fallback_rounds):
        try:
            out = json.load(open(os.path.join(d, name)))
            out["_dir"] = os.path.basename(d)
            return out
        except OSError:
            continue
    return None


def on(section):
    return not want or section in want


line = load("final_bench_detail.json") or load("final_bench.json")

if on("kernels"):
    for model in ("tiny", "medium", "large"):
        pre = "" if model == "tiny" else model + "_"
        ser, sq, pmc = load(pre + "serial_kernels.json"), load(pre + "sq_issue.json"), load(pre + "pmc_traffic.json")
        if not (ser and sq):
            continue
        if model == "tiny":
            alone = ((line or {}).get("roofline") or {}).get("serialised", {}).get("kernels", {})
        else:
            alone = (((line or {}).get("other_configs") or {}).get(model, {}).get("roofline") or {}).get("serialised_kernels", {})
        print(f"### {model}: kernels alone on the chip, image -> proof, batches of {ser['proofs_per_launch']} ({rnd}: {pre}serial_kernels.json, {pre}sq_issue.json, "
              f"{pre}pmc_traffic.json)\n")
        print("| kernel | µs per proof | share | launches per batch × avg µs | VALU issue util. | active / issue-stall / wait | waves / SIMD | "
              "counter HBM MB per launch | algorithmic GB/s alone (of 8 000) |")
        print("|---|---|---|---|---|---|---|---|---|")
        other = 0.0
        for k, v in ser["kernels"].items():
            if v["share"] < 0.004:
                other += v["us_per_proof"]
                continue
            q = sq["kernels"].get(k, {})
            a = alone.get(k, {})
            hb = (pmc or {}).get("kernels", {}).get(k, {}).get("hbm_bytes_per_launch")
            split = f"{q['active']:.2f} / {q['issue_stall']:.2f} / {q['wait']:.2f}" if "active" in q else ""
            print(f"| `{k}` | {v['us_per_proof']:.1f} | {100 * v['share']:.1f} % | {v['launches_per_batch']:.0f} × {v['avg_launch_us']:.0f} | "
                  f"{q.get('valu_issue_util', float('nan')):.2f} | {split} | {q.get('waves_per_simd', float('nan')):.2f} | "
                  + (f"{hb / 1e6:.0f} | " if hb else "— | ")
                  + (f"{a['algo_GBps']:.0f} ({a['frac_of_hbm_peak']:.3f}) |" if "algo_GBps" in a else "— |"))
        w = sq["whole_proof_serialised"]
        print(f"| everything else | {other:.1f} | | | | | | | |")
        print(f"| **whole proof, kernels one at a time** | **{ser['device_us_per_proof']:.0f}** | | | {w['valu_issue_util']:.2f} | | | | |\n")
        if pmc and "valu" in pmc:
            v = pmc["valu"]
            top = list(v["by_kernel"].items())[:6]
            print(f"VALU wave-instructions per proof ({model}): {v['wave_instructions_per_proof']:.4g} -- "
                  + ", ".join(f"`{k}` {100 * x / v['wave_instructions_per_proof']:.1f} %" for k, x in top) + ".\n")

if line and on("families"):
    r = line["roofline"]
    proofs = line["steps"] * line["proofs_per_step"]
    print(f"### kernel families, timed region of the committed line ({line['_dir']}/final_bench_detail.json; {line['ms_per_proof']:.4f} ms/proof)\n")
    print("| family | share of device time | algorithmic bytes per proof | basis | counter bytes per proof | counter ÷ algorithmic | algorithmic GB/s |")
    print("|---|---|---|---|---|---|---|")
    for k, f in r["families"].items():
        cb = f"{f['counter_bytes'] / proofs / 1e6:.0f} MB" if f.get("counter_bytes") else "—"
        ratio = f"{f['counter_over_algorithmic_bytes']:.1f}×" if f.get("counter_over_algorithmic_bytes") else "—"
        print(f"| {k} | {100 * f['share_of_device_time']:.1f} % | {f['algorithmic_bytes'] / proofs / 1e6:.1f} MB | {f['basis']} | {cb} | {ratio} | "
              f"{f['algo_GBps']:.0f} |")
    print(f"\nSURVEY 8d per proof: {line['algorithmic_bytes_per_proof'] / 1e6:.1f} MB by the formula, "
          f"{line['algorithmic_bytes_per_proof_charged'] / 1e6:.1f} MB charged by the library launch by launch.\n")

if line and on("models") and line.get("other_configs"):
    print(f"### the four configurations on one GPU ({line['_dir']}/final_bench_detail.json)\n")
    print("| model | k | provers × batch | ms / proof (image → proof) | proofs / hour | one prover alone, ms / proof | dominant kernel alone: frac of HBM peak | "
          "counter ÷ algorithmic (MSM family) | VALU: frac of 4-cycle issue rate | lone create_proof ms | lone image → proof ms | witness_run alone ms |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    rows = [("tiny", {"model": line["metric"].split(", ")[-1], "k": 14, "provers": line["provers_per_gpu"], "batch": line["batch"],
                      "ms_per_proof": line["ms_per_proof"], "proofs_per_hour": line["value"], "roofline": line["roofline"], "valu": line.get("valu"),
                      "lone": (line.get("lone") or {}).get("opted_in", {})})] + list(line["other_configs"].items())
    for m, o in rows:
        rf, lo = o.get("roofline") or {}, o.get("lone") or {}
        s_ = rf.get("serialised") or {}
        print(f"| {o['model'].replace('model_', '')} | {o['k']} | {o['provers']} × {o['batch']} | {o['ms_per_proof']:.3f} | {o['proofs_per_hour']:.3g} | "
              f"{s_.get('ms_per_proof', float('nan')):.3f} | `{rf.get('kernel')}` {s_.get('frac', float('nan')):.4f} | "
              + (f"{rf['msm_counter_over_algorithmic']:.1f}× | " if rf.get("msm_counter_over_algorithmic") else "— | ")
              + (f"{o['valu']['frac_of_four_cycle_issue_rate']:.2f} | " if o.get("valu") else "— | ")
              + f"{lo.get('create_proof_ms', float('nan')):.2f} | {lo.get('image_to_proof_ms', float('nan')):.2f} | {lo.get('witness_run_ms', float('nan')):.3f} |")
    print()

if line and on("lone") and line.get("lone"):
    print(f"### a lone proof of the tiny model, twice ({line['_dir']}/final_bench_detail.json: lone)\n")
    print("| | digit tables | gate | runtime settings | create_proof ms (median of 9) | image → proof ms (median of 9) | witness_run alone ms | same bytes in the plain order |")
    print("|---|---|---|---|---|---|---|---|")
    for form in ("opted_in", "default"):
        lo = line["lone"].get(form)
        if not lo:
            continue
        env = line["runtime_env"] if form == "opted_in" else lo.get("runtime_env", {})
        envs = ", ".join(f"{k}={v}" for k, v in env.items() if v is not None) or "none (as the runtime comes)"
        print(f"| {form.replace('_', '-')} | {lo['digit_table_bytes'] / 1e9:.1f} GB | {'on' if lo['gate'] else 'off'} | {envs} | {lo['create_proof_ms']:.3f} "
              f"({min(lo['create_proof_ms_samples']):.2f}–{max(lo['create_proof_ms_samples']):.2f}) | {lo.get('image_to_proof_ms', float('nan')):.3f} | "
              f"{lo.get('witness_run_ms', float('nan')):.3f} | {lo['bytes_equal_plain_order']} |")
    print()

sh = load("shard_compute_leg.json", ("r04",)) if on("shard") else None
if sh:
    print(f"### one rank's compute leg of a point-range-sharded step ({sh['_dir']}/shard_compute_leg.json; no collective; not a scaling curve)\n")
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
    print()

ms = load("msm_only_compute_leg.json") if on("msm") else None
if ms:
    print(f"### one rank's compute leg of --mode msm-only ({rnd}/msm_only_compute_leg.json; no collective; not a scaling curve)\n")
    print("| model | world | points per rank | digit width | MSMs per step | µs per MSM | × vs world 1 | MSMs / s of ONE rank | msm_accumulate ms / step | digits + hist + scan + scatter | "
          "heavy + strip + strip_sum |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in ms["rows"]:
        k = r["kernel_ms_per_step"]
        sort_ = sum(k.get(n, 0.0) for n in ("msm_digits", "msm_hist", "msm_scan", "msm_scatter"))
        red = sum(k.get(n, 0.0) for n in ("msm_heavy", "msm_strip", "msm_strip_sum"))
        print(f"| {r['model']} | {r['world']} | {r['points_per_rank']} | {r['digit_width']} | {r['msms_per_step']} | {r['us_per_msm']:.2f} | {r['x_vs_world_1']:.2f} | "
              f"{r['msms_per_s_one_rank']:.0f} | {k.get('msm_accumulate', 0):.2f} | {sort_:.2f} | {red:.2f} |")
    print()

wf = load("witness_run_forms.json") if on("witness") else None
if wf:
    print(f"### witness_run alone on the chip, µs, 1 image / batch of 16 ({rnd}/witness_run_forms.json)\n")
    print("| model | levels: recorded → parallel prefix | round 4's kernel, recorded program | + no scratch memory (operands in HBM) | + live values in LDS | "
          "parallel-prefix program, operands in HBM | parallel-prefix program, live values in LDS | LDS bytes | 8-byte / 32-byte cells | values in LDS / left in HBM | "
          "levels with a global barrier | 64-bit operations |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    by, rec = {}, {}
    for r in wf["rows"]:
        by.setdefault(r["model"], {})[r["ZG_WITNESS_LDS"]] = r
    for r in wf["rows_recorded_program"]:
        rec.setdefault(r["model"], {})[r["ZG_WITNESS_LDS"]] = r
    for m, d in by.items():
        h, l, rh, rl = d[0], d[-1], rec[m][0], rec[m][-1]
        old = wf["round4_kernel_us"][m]
        print(f"| {m} | {rl['levels']} → {l['levels']} | {old['1']:.0f} / {old['16']:.0f} | {rh['witness_run_us_1']:.0f} / {rh['witness_run_us_16']:.0f} | "
              f"{rl['witness_run_us_1']:.0f} / {rl['witness_run_us_16']:.0f} | {h['witness_run_us_1']:.0f} / {h['witness_run_us_16']:.0f} | "
              f"**{l['witness_run_us_1']:.0f} / {l['witness_run_us_16']:.0f}** | {l['lds_bytes']} | {l['narrow_cells']} / {l['wide_cells']} | "
              f"{l['values_in_lds']} / {l['values_in_hbm']} | {l['hbm_levels']} | {l['narrow_ops']} |")
    print()
