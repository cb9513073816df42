"""Every tuning knob of the product library (include/zg_halo2.h, "tuning") against the oracle.

A knob that silently yielded other bytes would be a parity hole a user could open with `export`: here each knob is set
to its non-default values IN THIS PROCESS (zg_tuning_set -- the environment variable of the same name is only its
starting value), a fresh prover is built under it, and proofs in both scheduling forms and in a lock-step batch must
equal the oracle's create_proof byte for byte; the stand-alone MSM entry is checked under the MSM knobs as well.
`test_every_knob_is_covered` fails when the library grows a knob this file does not walk, and
`test_the_library_reads_the_environment_in_one_place` (CPU) when a getenv appears outside the knob table."""
import os
import re

import numpy as np
import pytest

from circuits import toy_circuit, variant_circuit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# knob -> the non-default settings walked (the defaults are what every other test runs)
SETTINGS = {
    "ZG_MSM_C": [5, 10],
    "ZG_MSM_K": [4, 120],
    "ZG_MSM_K_LAT": [4, 48],
    "ZG_MSM_RB": [64, 128, 256],
    "ZG_MSM_LANES": [2, 4],
    "ZG_MSM_STRIP": [2, 4, 16],
    "ZG_MSM_NAF": [0, 3, 9, 16],
    "ZG_MSM_NAF_GL": [0, 5, 12],
    "ZG_MSM_RUNS": [0],
    "ZG_EVALH_GROUPED": [0],
    "ZG_EVALH9": [0],
    "ZG_SPLIT_DOMAIN": [0],
    "ZG_LAT_SPLIT_K": [0, 99],
    "ZG_LAT_FULL_C": [0, 4, 7],
    "ZG_LAT_FULL_K": [4, 48],
    "ZG_LAZY_DOT": [0],
    "ZG_WITNESS_LDS": [0],  # (shapes witness plans, not provers: walked with its own parity cases in tests/test_gpu_witness.py)
    "ZG_MSM_AFFINE": [0, 1, 2, 3, 4],
    "ZG_LAT_PULL": [0, 1],
    "ZG_LAT_GATE": [1],  # (engages from a prover's second proof in a form on: tests/test_gpu_prover.py walks it there)
    "ZG_MSM_HEAVY": [1, 5, 64],
    "ZG_MSM_TOPSPLIT": [0],
    "ZG_NTT9": [0, 1],  # (the default is one or the other by scheduling form)
}
# knobs that act together: walked as pairs as well
PAIRS = [({"ZG_MSM_RB": rb, "ZG_MSM_LANES": l}) for rb in (64, 128) for l in (2, 4)] + [
    {"ZG_MSM_NAF": 0, "ZG_MSM_NAF_GL": 0, "ZG_MSM_RUNS": 0},   # plain window tables everywhere
    {"ZG_EVALH9": 0, "ZG_EVALH_GROUPED": 0},
    {"ZG_MSM_C": 6, "ZG_MSM_NAF_GL": 0},
    {"ZG_LAT_SPLIT_K": 0, "ZG_SPLIT_DOMAIN": 0},
    {"ZG_LAT_FULL_C": 5, "ZG_MSM_RUNS": 0},
    {"ZG_MSM_AFFINE": 2, "ZG_MSM_NAF": 0, "ZG_MSM_NAF_GL": 0},   # affine rounds over the window tables
    {"ZG_MSM_AFFINE": 3, "ZG_MSM_K": 4},
    {"ZG_MSM_AFFINE": 1, "ZG_MSM_RUNS": 0},
    {"ZG_NTT9": 1, "ZG_SPLIT_DOMAIN": 0}, {"ZG_NTT9": 0, "ZG_EVALH9": 0}, {"ZG_NTT9": 1, "ZG_LAT_SPLIT_K": 0},
    {"ZG_MSM_TOPSPLIT": 0, "ZG_MSM_NAF": 9}, {"ZG_MSM_TOPSPLIT": 1, "ZG_MSM_NAF": 3}, {"ZG_MSM_TOPSPLIT": 1, "ZG_MSM_NAF": 16, "ZG_MSM_NAF_GL": 5},
]
CASES = [{k: v} for k, vs in SETTINGS.items() for v in vs] + PAIRS


def test_the_library_reads_the_environment_in_one_place():
    """(no GPU needed) the knob table in ctx.hip holds the library's only getenv"""
    hits = []
    for f in sorted(os.listdir(os.path.join(ROOT, "0g-halo2_amd", "csrc"))):
        if f.endswith((".hip", ".h", ".inc")):
            src = open(os.path.join(ROOT, "0g-halo2_amd", "csrc", f)).read()
            src = re.sub(r"//[^\n]*", "", src)
            hits += [(f, m.start()) for m in re.finditer(r"\bgetenv\s*\(", src)]
    assert [f for f, _ in hits] == ["ctx.hip"], hits


def test_every_knob_is_covered_and_documented(zg):
    names = zg.tuning_names()
    assert sorted(names) == sorted(SETTINGS), "a knob without a parity walk (or a walk without a knob)"
    header = open(os.path.join(ROOT, "include", "zg_halo2.h")).read()
    for n in names:
        assert n in header, f"{n} is not documented in include/zg_halo2.h"
    with pytest.raises(zg.ZgError):
        zg.tuning_set("ZG_NO_SUCH_KNOB", 1)
    for n in names:  # set / get / restore round trip, no GPU needed
        before = zg.tuning_get(n)
        zg.tuning_set(n, 7)
        assert zg.tuning_get(n) == 7
        zg.tuning_set(n, -5)
        assert zg.tuning_get(n) == -1
        zg.tuning_set(n, before)


@pytest.fixture(scope="module")
def workloads(orc):
    """circuits, witnesses, SRS and the oracle's proofs: independent of every knob, made once"""
    out = []
    for name, (cs, asg, ilen), k in (("toy k8 degree 6", toy_circuit(8, force_degree=6), 8),
                                     ("merged selectors k6", variant_circuit("merged_selectors", k=6), 6),
                                     ("wide lookup k6", variant_circuit("wide_lookup", k=6), 6)):
        img = cs.to_c()
        params = orc.params_new(k, 0xABCDEF)
        vk_repr = orc.fr_from_int(0x1234567)
        fixed, sigma = asg.fixed_values(), asg.sigma_values()
        pk = orc.ProvingKey(img, fixed, sigma, params, vk_repr)
        adv, inst = asg.advice_values(), asg.instance_values(ilen)
        want = {}
        for seed in (1, 2, 3, 4):
            st, proof, _ = orc.create_proof(pk, adv, inst, seed)
            assert st == 0
            want[seed] = proof
        assert orc.verify_proof_pairing(pk, inst, want[1]) == 1
        out.append(dict(name=name, img=img, fixed=fixed, sigma=sigma, g=params.g_np(), gl=params.g_lagrange_np(),
                        vk_repr=vk_repr, adv=adv, inst=inst, want=want))
    prm = orc.params_new(10)
    g = prm.g_np()
    scal = np.stack([orc.fill_fr(77, 1 << 10), orc.fill_fr_sparse(78, 1 << 10)])
    msm_want = [orc.msm(s, g, threads=4) for s in scal]
    return out, (g, scal, msm_want)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: ",".join(f"{k}={v}" for k, v in c.items()))
def test_knob_settings_give_the_oracles_bytes(ctx, zg, orc, workloads, case):
    circuits, (g, scal, msm_want) = workloads
    before = {k: zg.tuning_get(k) for k in case}
    try:
        for k, v in case.items():
            zg.tuning_set(k, v)
        for w in circuits:
            prover = zg.Prover(ctx, w["img"], w["fixed"], w["sigma"], w["g"], w["gl"], w["vk_repr"])
            prover.set_overlap(True)    # latency form: side stream, single coset, several lanes per EC addition
            assert prover.prove(w["adv"], w["inst"], 1) == w["want"][1], (case, w["name"], "latency form")
            prover.set_overlap("tables")  # ... and over its digit tables (ZG_LAT_FULL_C / _K act here; explicit since round 4)
            assert prover.prove(w["adv"], w["inst"], 1) == w["want"][1], (case, w["name"], "latency form, digit tables")
            prover.set_overlap(False)   # throughput form: split domain, strip reduction, free-position digits
            assert prover.prove(w["adv"], w["inst"], 2) == w["want"][2], (case, w["name"], "throughput form")
            prover.set_batch(3)
            got, sts = prover.prove_batch([w["adv"]] * 3, [w["inst"]] * 3, [2, 3, 4])
            assert sts == [0, 0, 0] and got == [w["want"][s] for s in (2, 3, 4)], (case, w["name"], "batch of three")
            prover.close()
        # the stand-alone MSM entries under the same knobs: window form (latency), then the free-position form
        bases = ctx.register_bases(g)
        for latency in (True, False):
            ctx.set_msm_latency(latency)
            if not latency:
                ctx.enable_bit_table(bases, 9)
            got = ctx.msm_batch(bases, scal)
            for b in range(len(scal)):
                assert np.array_equal(got[b], msm_want[b]), (case, "zg_msm_batch", latency, b)
        ctx.set_msm_latency(True)
        bases.free()
    finally:
        for k, v in before.items():
            zg.tuning_set(k, v)
