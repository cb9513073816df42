#!/usr/bin/env python3
"""The transforms of one lock-step batch alone, per butterfly back end (run on the GPU box; under rocprofv3 for counters):
    python tools/ntt_forms.py [k] [arrays] [repeats]
`arrays` columns of 2^k values: the inverse transform (values -> coefficients) and the extension to the 4n-point coset."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import zg_halo2 as zg  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 14
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 192
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n, ek = 1 << k, k + 2
en = 1 << ek
ctx = zg.Ctx(0)
gen = np.random.default_rng(1)
a = gen.integers(0, 1 << 62, size=(batch, n, 4), dtype=np.int64).astype(np.uint64)
a[..., 3] &= np.uint64((1 << 60) - 1)
a = torch.from_numpy(a.view(np.int64)).cuda()
out = torch.empty((batch, en, 4), dtype=torch.int64, device="cuda")
om, omi = zg.domain_omega(k)
div = zg.fr_from_int(pow(n, -1, zg.FR_MODULUS))
forms = [int(x) for x in os.environ.get("FORMS", "0,1").split(",")]
for form in forms:
    zg.tuning_set("ZG_NTT9", form)
    for _ in range(2):
        ctx.ntt_batch_dev(a.data_ptr(), n, batch, k, omi, div)
        ctx.coeff_to_extended_batch_dev(a.data_ptr(), n, out.data_ptr(), en, batch, k, ek)
    ctx.sync()
    ctx.profile(True)
    for _ in range(reps):
        ctx.ntt_batch_dev(a.data_ptr(), n, batch, k, omi, div)
    st1 = ctx.profile_collect()
    for _ in range(reps):
        ctx.coeff_to_extended_batch_dev(a.data_ptr(), n, out.data_ptr(), en, batch, k, ek)
    st2 = ctx.profile_collect()
    ctx.profile(False)
    print(f"ZG_NTT9={form} k={k} arrays={batch}: values->coefficients {dict((x, round(y[1] / reps * 1e3, 1)) for x, y in st1.items())} us | "
          f"n->4n coset {dict((x, round(y[1] / reps * 1e3, 1)) for x, y in st2.items())} us", flush=True)
