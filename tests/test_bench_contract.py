"""The bench line's contract.  Round 4's line was 20.7 KB and the harness could not parse it; since round 5 stdout carries ONE
compact line built by bench.format_line (< 4 KB) and everything else goes to bench_detail.json.  Checked here without a GPU:
the line built from a deliberately oversized record through the SAME function main() uses; the committed line and detail
record of the final tree (profiles/r05/final_bench_line.json, final_bench_detail.json) with the accounting rules of round 4
(no figure above the HBM peak, the dominant kernel picked from the serialised pass, a multi-kernel unit charged once,
SURVEY's per-proof bytes equal to what the library charged); that a failing tail leg cannot lose the headline; the launcher
decision of `--gpus N`, eight ranks of plumbing on the CPU (--dry-run), the runtime settings an N > 1 run does NOT get; and
that nothing of the timed path imports the oracle (it is the checker and the CPU baseline leg only)."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r05", "final_bench_line.json")
DETAIL = os.path.join(ROOT, "profiles", "r05", "final_bench_detail.json")
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")


def _bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("zg_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _oversized_record():
    """what main() hands format_line after every tail leg, with every free-text and table field far larger than it ever is"""
    kernels = {f"kernel_{i:02d}": {"avg_launch_ms": 1.23456789 * i, "share_of_device_time": 0.01, "algo_GBps": 123.456, "frac_of_hbm_peak": 0.0154321,
                                   "hbm_bytes_per_launch": 123456789 * i} for i in range(80)}
    fam = {f: {"share_of_device_time": 0.2, "algorithmic_bytes": 1e9, "basis": "x" * 60, "algo_GBps": 100.0, "frac_of_hbm_peak": 0.0125,
               "counter_bytes": 2e9, "counter_over_algorithmic_bytes": 15.3} for f in ("msm", "ntt", "evaluate_h", "sort", "products", "openings", "witness")}
    lone = {"create_proof_ms": 2.1012345, "image_to_proof_ms": 3.4012345, "witness_run_ms": 1.2012345, "witness_finish_ms": 0.02, "digit_table_bytes": 77309411328,
            "gate": True, "bytes_equal_plain_order": True, "create_proof_ms_samples": [2.1] * 9, "image_to_proof_ms_samples": [3.4] * 9,
            "phase_ms": {p: 0.3 for p in ("advice", "lookups_permuted", "products", "h", "evals", "gwc", "total", "host_sort")}}
    return {
        "metric": "create_proof proofs/hour, model_28input_256entry_1hash_1bpi", "value": 5432109.87654321, "unit": "proofs/hour", "n_gpus": 1,
        "steps": 20, "warmup": 5, "ms_per_step": 254.123456789, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32x8 (254-bit Montgomery integers)", "data": "checked-in model " + "d" * 500,
        "config": {"workload": "image -> proof " + "w" * 2000, "class_scores": list(range(10)), "proofs_per_step": 384, "parallelism": "p" * 900},
        "mode": "single-gpu", "timed_region": "image_to_proof", "ms_per_proof": 0.66177983, "verified": True, "detail": "v" * 400,
        "collective": {"backend": None, "ranks_seen": None, "how": "h" * 300},
        "from_resident_columns_ms_per_proof": 0.6512345, "create_proof_wall_s": 0.0021012345, "image_to_proof_wall_s": 0.0034012345,
        "lone": {"opted_in": lone, "default": dict(lone, gate=False, digit_table_bytes=0, runtime_env={}), "opted_in_is": "o" * 300},
        "roofline": {"bound": "hbm", "kernel": "msm_accumulate", "kernel_picked_by": "k" * 200, "achieved": 50.123456, "peak": 8000.0, "unit": "GB/s",
                     "frac": 50.123456 / 8000.0, "traffic": 1973334731, "avg_launch_ms": 6.0254321, "algo_bytes_per_launch": 302.0e6,
                     "msm_counter_over_algorithmic": 15.3, "families": fam, "durations": "d" * 300, "note": "n" * 300,
                     "serialised": {"frac": 0.0220123, "avg_launch_ms": 1.7132821, "ms_per_proof": 0.9381234, "kernels": kernels, "families": fam,
                                    "note": "s" * 300}},
        "valu": {"frac_of_four_cycle_issue_rate": 0.9112345, "per_kernel_alone": {k_: 0.5 for k_ in kernels}, "peak_note": "q" * 300},
        "kernels": kernels,
        "cpu_baseline": {"value": 2871.123, "unit": "proofs/hour", "cores": 16, "kind": "port", "wall_s": 1.2538, "sample": "s" * 700,
                         "sample_long": "t" * 900, "samples_s": [1.25] * 7, "phase_ms": {"h": 915.5}},
        "other_configs": {m: {"model": m * 10, "k": 15, "ms_per_proof": 1.3812345, "image_to_proof_wall_s": 0.0051234, "verified": True,
                              "roofline": {"kernel": "ntt_cols", "serialised_kernels": kernels}, "lone": lone} for m in ("small", "medium", "large")},
        "algorithmic_bytes_per_proof": 391.6e6, "runtime_env": {"GPU_MAX_HW_QUEUES": "16", "HIP_FORCE_DEV_KERNARG": "1", "HSA_ENABLE_INTERRUPT": "0"},
        "errors": {"other_large": {"error": "e" * 500, "traceback": "t" * 1500}}, "detail_file": "bench_detail.json",
    }


def test_the_line_is_compact_whatever_the_record_holds():
    """built through the function main() uses; < 4 KB; json round trip; contract keys, roofline and
    cpu_baseline present; the optional keys the verdict lists."""
    b = _bench()
    rec = _oversized_record()
    assert len(json.dumps(rec)) > 20000  # (the record itself is of round 4's size)
    line = b.format_line(rec)
    assert "\n" not in line and len(line.encode()) < 4096 == b.LINE_LIMIT
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert set(d["config"]) == {"workload", "proofs_per_step", "parallelism"} and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algo_bytes_per_launch", "avg_launch_ms"):
        assert k in r, k
    assert set(r["serialised"]) == {"frac", "avg_launch_ms", "ms_per_proof"} and r["msm_counter_over_algorithmic"] == 15.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 * r["frac"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "wall_s"):
        assert k in c, k
    assert len(c["sample"]) <= 120
    assert d["verified"] is True and d["ms_per_proof"] == pytest.approx(0.66178, rel=1e-5)
    assert d["create_proof_wall_s"] and d["image_to_proof_wall_s"] and d["from_resident_columns_ms_per_proof"]
    for form in ("opted_in", "default"):
        assert {"create_proof_ms", "image_to_proof_ms", "witness_run_ms", "gate"} <= set(d["lone"][form])
    assert d["lone"]["default"]["gate"] is False and d["lone"]["opted_in"]["digit_table_bytes"] > 0
    assert set(d["other_configs"]) == {"small", "medium", "large"} and d["other_configs"]["medium"]["ms_per_proof"] > 0
    assert d["valu"] == {"frac_of_four_cycle_issue_rate": pytest.approx(0.911235, rel=1e-5)}
    assert d["errors"] == ["other_large"] and d["detail"] == "bench_detail.json"
    for heavy in ("kernels", "families", "per_kernel_alone", "samples_s"):
        assert heavy not in line, heavy


def test_a_line_that_still_does_not_fit_sheds_optional_keys_not_the_contract():
    b = _bench()
    rec = _oversized_record()
    line = b.format_line(rec, limit=2400)
    d = json.loads(line)
    assert len(line.encode()) < 2400
    for k in CONTRACT:
        assert k in d, k
    assert "roofline" in d and "cpu_baseline" in d  # (the last to go)


def test_a_failing_tail_leg_cannot_lose_the_headline(capfd):
    """the headline goes to stderr and to the detail file when the timed region ends; a leg that raises
    is named in `errors`; stdout gets the ONE line exactly once, also through the atexit path."""
    b = _bench()
    b.DETAIL = os.path.join(ROOT, "tests", "_bench_detail_test.json")
    try:
        em = b.Emitter()
        rec = {k_: v for k_, v in _oversized_record().items() if k_ not in ("errors", "cpu_baseline", "other_configs")}
        em.early(rec)
        assert json.load(open(b.DETAIL))["value"] == rec["value"]

        def boom():
            raise MemoryError("hipMalloc failed at k = 17")

        em.leg("other_large", boom)
        em.leg("cpu_baseline", lambda: {"cpu_baseline": {"value": 1.0, "unit": "proofs/hour", "cores": 16, "kind": "port", "sample": "x", "wall_s": 1.0}})
        em.final()
        em.final()  # (atexit calls it again: nothing more is printed)
        out, err = capfd.readouterr()
        lines = [ln for ln in out.splitlines() if ln.strip()]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["errors"] == ["other_large"] and d["cpu_baseline"]["value"] == 1.0 and d["value"] == pytest.approx(rec["value"], rel=1e-5)
        assert "headline" in err and '"metric"' in err and "other_large" in err
        full = json.load(open(b.DETAIL))
        assert "hipMalloc failed" in full["errors"]["other_large"]["error"] and "leg_seconds" in full
    finally:
        for f in (b.DETAIL, b.DETAIL + ".tmp"):
            if os.path.exists(f):
                os.remove(f)


@pytest.mark.skipif(not os.path.exists(LINE), reason="no committed round-5 line yet")
def test_committed_bench_line_has_the_contracts_keys():
    raw = open(LINE).read().strip()
    assert "\n" not in raw and len(raw.encode()) < 4096
    d = json.loads(raw)
    for k in CONTRACT + ("roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "proofs/hour" and d["higher_is_better"] is True and d["n_gpus"] == 1
    assert d["scaling"] == "weak" and d["vs_baseline"] is None  # (BASELINE.md publishes no number for this metric)
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "model_28input_256entry_1hash_1bpi" in d["metric"]
    # the timed region is the reference's: image -> proof (benches/bench.rs:35 times wnn.proof)
    assert d["config"]["workload"].startswith("image -> proof")
    assert d["from_resident_columns_ms_per_proof"] > 0
    # value is the whole job's rate over the timed region: steps x proofs per step / time
    expect = d["steps"] * d["config"]["proofs_per_step"] / (d["ms_per_step"] * d["steps"] / 1e3) * 3600.0
    assert abs(d["value"] - expect) < 1e-4 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 * r["frac"]
    # achieved = algorithmic bytes per launch / average launch duration of the dominant kernel
    assert abs(r["achieved"] - r["algo_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-4 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > r["algo_bytes_per_launch"] * 0.5
    s = r["serialised"]
    assert 0 < s["frac"] < 1 and s["frac"] >= r["frac"]  # (alone on the chip: never slower)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1
    assert "SLOWER than real halo2" in c["sample"]  # (the ratio to it is not a result: VERDICT r3 weak 10)
    assert d["verified"] is True and "errors" not in d
    # the lone proof, twice: opted-in (digit tables + gate + runtime settings) and as the library comes
    lo, ld = d["lone"]["opted_in"], d["lone"]["default"]
    assert lo["digit_table_bytes"] > 0 and lo["gate"] is True and lo["bytes_equal_plain_order"] is True
    assert ld["digit_table_bytes"] == 0 and ld["gate"] is False
    assert lo["create_proof_ms"] < ld["create_proof_ms"] and lo["image_to_proof_ms"] > lo["create_proof_ms"]
    assert abs(d["create_proof_wall_s"] * 1e3 - lo["create_proof_ms"]) < 1e-3
    assert set(d["other_configs"]) == {"small", "medium", "large"}


@pytest.mark.skipif(not os.path.exists(DETAIL), reason="no committed round-5 detail record yet")
def test_roofline_accounting_can_be_true():
    """(a) the dominant kernel is the top of the SERIALISED pass, (b) a multi-kernel unit is charged once
    -- the MSM family's algorithmic bytes are ONE n * 96 + 96 per MSM while its counter bytes are the sum over its kernels
    -- (c) no kernel and no family moves bytes faster than the HBM peak, (d) SURVEY's per-proof figure == what the library
    charged launch by launch."""
    d = json.load(open(DETAIL))
    r = d["roofline"]
    ser = r["serialised"]
    top = max(ser["kernels"].items(), key=lambda kv: kv[1]["share_of_device_time"])[0]
    assert r["kernel"] == ser["kernel"] == top
    assert "serialised" in r["kernel_picked_by"]
    for name, k in ser["kernels"].items():
        assert 0 <= k["frac_of_hbm_peak"] <= 1.0, (name, k)
    for name, k in d["kernels"].items():
        assert 0 <= k["frac_of_hbm_peak"] <= 1.0, (name, k)
    for fams in (r["families"], ser["families"]):
        for name, f in fams.items():
            assert 0 <= f["frac_of_hbm_peak"] <= 1.0, (name, f)
    # the library's charges against SURVEY 8d's formula (bench.algorithmic_bytes_per_proof)
    assert abs(d["algorithmic_bytes_per_proof"] - d["algorithmic_bytes_per_proof_charged"]) < 1e-6 * d["algorithmic_bytes_per_proof"]
    proofs = d["steps"] * d["proofs_per_step"]
    unit_fams = [f for f in r["families"].values() if f["basis"].startswith("SURVEY")]
    assert abs(sum(f["algorithmic_bytes"] for f in unit_fams) / proofs - d["algorithmic_bytes_per_proof"]) < 1e-6 * d["algorithmic_bytes_per_proof"]
    # the MSM as the unit SURVEY defines: 30 commitments of n * 96 + 96 bytes per proof, once
    n = 1 << 14
    assert abs(r["families"]["msm"]["algorithmic_bytes"] / proofs - 30 * (n * 96 + 96)) < 1.0
    # ... against the counter bytes of ALL its kernels (round 3 divided them by an eightfold algorithmic figure)
    msm = r["families"]["msm"]
    assert msm["counter_over_algorithmic_bytes"] is None or msm["counter_over_algorithmic_bytes"] > 8.0
    assert msm["streamed_bytes_of_its_kernels"] > msm["algorithmic_bytes"]
    # the witness program's kernels are in the timed region (the reference's region) and are a family of their own
    assert "witness" in r["families"] and d["timed_region"] == "image_to_proof"
    # every other model carries its own roofline record
    for m in ("small", "medium", "large"):
        o = d["other_configs"][m]
        assert o["ms_per_proof"] > 0 and o["verified"] is True, m
        assert 0 < o["roofline"]["serialised"]["frac"] < 1 and o["roofline"]["kernel"], m


def test_bench_touches_the_oracle_only_as_checker_and_cpu_baseline():
    src = open(os.path.join(ROOT, "bench.py")).read()
    # every import of the oracle's binding sits inside a checking / CPU-baseline function
    for m in re.finditer(r"^\s*import orc\b", src, re.M):
        head = src[:m.start()]
        fn = re.findall(r"^def (\w+)\(", head, re.M)[-1]
        assert fn in ("cpu_baseline", "_oracle_pk", "msm_only_check", "msm_only_cpu_baseline"), fn
    callers = [fn for fn in re.findall(r"^def (\w+)\(", src, re.M) if "_oracle_pk(" in src.split(f"def {fn}(")[1].split("\ndef ")[0]]
    assert callers == ["verify_last_step"], callers
    for timed in ("measure", "measure_headline", "run_steps", "lone_probe", "serialised_pass"):
        body = src.split(f"def {timed}(")[1].split("\ndef ")[0]
        assert "orc" not in re.findall(r"\b\w+\b", body), timed  # (the timed regions themselves)
    step = src.split("    def step():")[1].split("\n    def ")[0]  # msm-only's timed step
    assert "orc" not in re.findall(r"\b\w+\b", step)


def test_runtime_settings_are_made_for_a_single_gpu_process_only():
    """HIP_FORCE_DEV_KERNARG and HSA_ENABLE_INTERRUPT=0 help only the lone-proof probe and have
    never run beside RCCL: a rank of an N > 1 run and the launcher (whose environment the ranks inherit) do not get them."""
    b = _bench()
    one = b.runtime_env_for(["--steps", "20"], {})
    assert one == {"GPU_MAX_HW_QUEUES": "16", "HIP_FORCE_DEV_KERNARG": "1", "HSA_ENABLE_INTERRUPT": "0"}
    assert b.runtime_env_for(["--gpus", "1"], {"WORLD_SIZE": "1"}) == one
    for argv, env in ((["--gpus", "8"], {}), (["--gpus=2"], {}), (["--gpus", "8"], {"WORLD_SIZE": "8", "RANK": "3"}), ([], {"WORLD_SIZE": "2"})):
        assert b.runtime_env_for(argv, env) == {"GPU_MAX_HW_QUEUES": "16"}, (argv, env)
    assert b.runtime_env_for([], {"ZG_BENCH_PLAIN_ENV": "1"}) == {}


def test_gpus_n_starts_n_ranks_as_a_child():
    """`python bench.py --gpus N` (the driver's command shape) must measure N GPUs: argv + environment
    in, the child's command out -- decided before any GPU call."""
    b = _bench()
    argv = ["--gpus", "4", "--steps", "5", "--warmup", "2"]
    cmd = b.launcher_command(argv, 4, {"ZG_BENCH_PORT": "29999"})
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv  # (same arguments, --gpus included)
    # no port given: a free one is drawn
    port = int(b.launcher_command(argv, 4, {})[b.launcher_command(argv, 4, {}).index("--master-port") + 1])
    assert 1024 < port < 65536
    # a rank started by torch.distributed.run runs the bench itself ...
    assert b.launcher_command(argv, 4, {"WORLD_SIZE": "4", "RANK": "1"}) is None
    # ... one GPU needs no launcher ...
    assert b.launcher_command(["--steps", "5"], 1, {}) is None
    assert b.launcher_command(["--gpus", "1"], 1, {"WORLD_SIZE": "1"}) is None
    # ... and a world that contradicts --gpus is refused, never printed as an N-GPU line
    with pytest.raises(SystemExit):
        b.launcher_command(argv, 4, {"WORLD_SIZE": "2"})
    with pytest.raises(SystemExit):
        b.launcher_command(["--gpus", "1"], 1, {"WORLD_SIZE": "8"})


def test_launcher_is_decided_before_the_first_gpu_call_and_never_execs():
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src.split("def main():")[1]
    assert main.index("launcher_command(") < main.index("setup_ranks(")
    ranks = src.split("def setup_ranks(")[1].split("\ndef ")[0]
    assert ranks.index("torch.cuda.is_available()") < ranks.index("torch.cuda.set_device")
    # nothing at import time or before the launcher decision touches the GPU
    head = src.split("def main():")[0]
    top_level = "\n".join(ln for ln in head.splitlines() if ln and not ln.startswith((" ", "#", "def ", "class ", '"""')))
    assert "torch.cuda" not in top_level and "zg.Ctx" not in top_level
    assert "os.exec" not in src.replace("never os.exec*", "")


def test_relay_passes_the_json_line_and_the_exit_code(capfd):
    b = _bench()
    rc = b.relay([sys.executable, "-c", "import sys; print('banner'); print('{\"metric\": \"m\", \"n_gpus\": 2}'); sys.exit(0)"])
    out, err = capfd.readouterr()
    assert rc == 0 and out.strip() == '{"metric": "m", "n_gpus": 2}' and "banner" in err
    rc = b.relay([sys.executable, "-c", "import sys; sys.exit(7)"])
    assert rc == 7
    rc = b.relay([sys.executable, "-c", "print('no line')"])
    assert rc != 0  # (ranks that exit 0 without a line did not measure anything)
    capfd.readouterr()


@pytest.mark.parametrize("mode", ["replicas", "shard-msm"])
def test_eight_ranks_of_plumbing_without_a_gpu(mode):
    """the driver's shape is 8 ranks; a one-GPU box allows six GPU processes, so the 8-rank bring-up --
    launcher, rendezvous on 127.0.0.1, group creation order (one exchange group per prover in shard-msm), ranks counted by the
    backend, barriers, MAX over ranks, ONE line -- is rehearsed on the CPU with --dry-run (no proof is made; the line says so)."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--mode", mode, "--provers", "2",
                        "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["collective"] == {"backend": "gloo", "ranks_seen": 8}
    assert d["value"] == 0.0 and "DRY RUN" in d["metric"] and d["mode"] == mode


def test_the_power_sampler_reads_hwmon_files_and_is_silent_without_them(tmp_path, monkeypatch):
    """bench.PowerSampler against a made-up hwmon tree (two cards, one idle), and on a box without any"""
    import glob as glob_mod
    import time

    bench = _bench()
    for card, watts, hz in (("card3", 1300e6, 2250e6), ("card4", 240e6, 150e6)):
        d = tmp_path / "drm" / card / "device" / "hwmon" / "hwmon7"
        d.mkdir(parents=True)
        (d / "power1_input").write_text(f"{int(watts)}\n")
        (d / "freq1_input").write_text(f"{int(hz)}\n")
        (d / "power1_cap").write_text("1400000000\n")
    real_glob = glob_mod.glob
    monkeypatch.setattr(glob_mod, "glob", lambda pat: real_glob(pat.replace("/sys/class/drm", str(tmp_path / "drm"))))
    s = bench.PowerSampler(period=0.005).start()
    time.sleep(0.05)
    got = s.stop()
    assert [c["card"] for c in got["cards"]] == ["card3"] and all(c["power_w_avg"] == 1300.0 and c["sclk_mhz_avg"] == 2250.0 and c["power_cap_w"] == 1400.0
                                                       for c in got["cards"]), got  # (the idle card is not reported)
    assert len(got["cards"]) == 1 and got["cards"][0]["samples"] >= 2 and got["matched_by_pci_address"] is False
    monkeypatch.setattr(glob_mod, "glob", lambda pat: [])
    assert bench.PowerSampler().start().stop() == {}
    line = json.loads(bench.format_line({**_oversized_record(), "power": {**got, "joules_per_proof": 0.8571234}}))
    assert line["power"]["power_w_avg"] == 1300.0 and line["power"]["joules_per_proof"] == 0.857123
