#!/usr/bin/env python3
"""Time the oracle's create_proof on this machine's cores (no GPU):   python tools/oracle_time.py [model] [threads] [repeats]
Prints the bench's cpu_baseline object and the sha256 of the first proof (to see that a change of the oracle kept its bytes)."""
import hashlib
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (puts harness/, oracle/ and the package on the path)

model = sys.argv[1] if len(sys.argv) > 1 else "tiny"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 3
k, name = bench.MODELS[model]
wnn = bench.wnn_model.synthetic_wnn() if model == "large" else bench.wnn_model.load_checked_in(name)
cs, asg, ilen, _ = bench.wnn_circuit.build(wnn, bench.wnn_model.load_test_image(), k, compress_selectors=True)
c = types.SimpleNamespace(k=k, img=cs.to_c(), fixed=asg.fixed_values(), sigma=asg.sigma_values(), advice=asg.advice_values(),
                          instance=asg.instance_values(ilen), vk_repr=bench.np.array(bench.limbs(0xC0FFEE * bench.MONT % bench.R), dtype=bench.np.uint64))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402

orc.load().orc_set_threads(threads)
params = orc.params_new(c.k, 0x5EED)
pk = orc.ProvingKey(c.img, c.fixed, c.sigma, params, c.vk_repr)
st, proof, _ = orc.create_proof(pk, c.advice, c.instance, 1)
print("status", st, "bytes", len(proof), "sha256", hashlib.sha256(proof).hexdigest())
out = bench.cpu_baseline(c, len(proof), threads, repeats)
print(json.dumps({k: out[k] for k in ("wall_s", "samples_s", "phase_ms")}))
