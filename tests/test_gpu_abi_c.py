"""The boundary driven from plain C (tests/abi/c_driver.c: zg_params_new, zg_bases_register + zg_msm, zg_prover_create,
zg_prover_prove, zg_prover_prove_batch -- no Python in that process, the call shape of a Rust `extern "C"` binding):
its MSM result and its proof bytes must be the oracle's for the statement of tests/abi/c_driver_data.h."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_driver_results_equal_the_oracle(orc, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests", "abi"))
    import gen_c_driver_data as gen
    from circuits import variant_circuit

    exe = str(tmp_path / "c_driver")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I",
                    os.path.join(ROOT, "tests", "abi"), os.path.join(ROOT, "tests", "abi", "c_driver.c"), "-L",
                    os.path.join(ROOT, "0g-halo2_amd"), "-lzg_halo2", "-Wl,-rpath," + os.path.join(ROOT, "0g-halo2_amd"), "-o", exe],
                   check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = dict(line.split(" ", 1) for line in r.stdout.strip().splitlines())
    assert "gfx950" in got["version"]

    cs, asg, ilen = variant_circuit("no_lookup", k=gen.K)
    s = orc.fr_from_int(gen.S_INT)
    params = orc.params_from_scalar(gen.K, s)
    pk = orc.ProvingKey(cs.to_c(), asg.fixed_values(), asg.sigma_values(), params, orc.fr_from_int(gen.VK_INT))
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    want_msm = orc.msm(adv[0], params.g_lagrange_np())
    assert bytes.fromhex(got["msm"].strip()) == np.ascontiguousarray(want_msm).tobytes(), "zg_msm from C"
    for label, key in (("proof", gen.KEYS[0]), ("batch0", gen.KEYS[1]), ("batch1", gen.KEYS[2])):
        st, want, _ = orc.create_proof(pk, adv, inst, key)
        assert st == 0 and bytes.fromhex(got[label].strip()) == want, label
    assert orc.verify_proof_pairing(pk, inst, bytes.fromhex(got["proof"].strip())) == 1
