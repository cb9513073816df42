"""The benchmarked shape's parity inside `pytest -m gpu`: what bench.py times -- model_28input_256entry_1hash_1bpi at
k = 14, 12 provers forked from one resident proving key, lock-step batches of 32, one host thread per prover, the
bit-position tables of g / g_lagrange (0.27 GB each: out of the Infinity Cache) and 12 x 32 proof slots in play -- with
EVERY proof checked: each is made twice, by another prover, in another slot, at another step, and the twins must agree
byte for byte; one proof of every prover is compared with the oracle's create_proof (tools/soak.py, reduced to 2 steps).
The second test does the same from image bytes (witness program on the device under concurrent provers)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_bench_shape_12_provers_x_32_slots_twins_and_oracle():
    import soak

    r = soak.soak(12, 32, 2, images=False, oracle_per_prover=1, log=lambda *_: None)
    assert r["proofs"] == 768 and r["twin_pairs"] == 384
    assert r["twin_mismatches"] == 0
    assert r["oracle_checked"] >= 12 and r["oracle_mismatches"] == 0


def test_image_to_proof_under_concurrent_provers_twins_and_oracle():
    import soak

    r = soak.soak(6, 8, 2, images=True, oracle_per_prover=1, log=lambda *_: None)
    assert r["proofs"] == 96 and r["twin_mismatches"] == 0
    assert r["oracle_checked"] >= 6 and r["oracle_mismatches"] == 0
