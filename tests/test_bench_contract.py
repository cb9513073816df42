"""The bench line's contract, checked on the committed line of the final tree (profiles/r03/final_bench.json) and on
bench.py's own source: the keys the driver parses, the roofline and cpu_baseline objects, and that nothing of the timed
path imports the oracle (it is the checker and the CPU baseline leg only).  No GPU needed."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contracts_keys():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03", "final_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "proofs/hour" and d["higher_is_better"] is True and d["n_gpus"] == 1
    assert d["scaling"] == "weak" and d["vs_baseline"] is None  # (BASELINE.md publishes no number for this metric)
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "model_28input_256entry_1hash_1bpi" in d["metric"]
    # value is the whole job's rate over the timed region: steps x proofs per step / time
    assert abs(d["value"] - d["steps"] * d["proofs_per_step"] / (d["ms_per_step"] * d["steps"] / 1e3) * 3600.0) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / average launch duration of the dominant kernel
    assert abs(r["achieved"] - r["algo_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > r["algo_bytes_per_launch"] * 0.5
    s = r["serialised"]
    assert s["kernel"] == r["kernel"] and 0 < s["frac"] < 1 and s["frac"] >= r["frac"]  # (alone on the chip: never slower)
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1
    assert d["verified"] is True


def test_bench_touches_the_oracle_only_as_checker_and_cpu_baseline():
    src = open(os.path.join(ROOT, "bench.py")).read()
    # every import of the oracle's binding sits inside verify_last_step / image_to_proof's check / cpu_baseline
    for m in re.finditer(r"^\s*import orc\b", src, re.M):
        head = src[:m.start()]
        fn = re.findall(r"^def (\w+)\(", head, re.M)[-1]
        assert fn in ("cpu_baseline", "verify_last_step", "image_to_proof", "verify_proofs"), fn
    assert "import orc" not in src.split("def measure(")[1].split("\ndef ")[0]  # (the timed region itself)
