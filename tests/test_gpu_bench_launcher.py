"""`python bench.py --gpus 2` end to end on the GPU box -- the driver's command shape: the process that
parses the arguments starts its two ranks as a child torch.distributed.run, relays ONE JSON line, and that line says two
GPUs and two ranks counted by the collective backend.  This pool has one GPU per box, so both ranks share cuda:0 and the
backend is gloo (RCCL refuses two ranks on one device): a rehearsal of the plumbing, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["replicas", "shard-msm"])
def test_gpus_2_starts_two_ranks_and_says_so(mode):
    env = dict(os.environ, ZG_BENCH_DEVICE="0", ZG_BENCH_BACKEND="gloo", ZG_BENCH_STALL_S="240")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", mode, "--steps", "1", "--warmup", "1",
           "--provers", "2", "--batch", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # the contract: ONE json line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["collective"]["ranks_seen"] == 2 and d["collective"]["backend"] == "gloo"
    assert d["rccl_ranks"] is None and d["ranks_share_a_device"] is True
    assert d["verified"] is True
    assert d["mode"] == mode and d["scaling"] == ("weak" if mode == "replicas" else "strong")
    # replicas: 2 ranks x 2 provers x 4 proofs per step; shard-msm: both ranks prove the SAME 2 x 4 proofs
    assert d["proofs_per_step"] == 8
    expect = 8 * (2 if mode == "replicas" else 1) / (d["ms_per_step"] / 1e3) * 3600.0
    assert abs(d["value"] - expect) < 1e-6 * expect


@pytest.mark.gpu
def test_a_world_that_contradicts_gpus_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
