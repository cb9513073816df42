"""Per-kernel MSM timing on characteristic scalar distributions (tuning aid; run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
import numpy as np
import zg_halo2 as zg

R = zg.FR_MODULUS
k = int(sys.argv[1]) if len(sys.argv) > 1 else 14
n = 1 << k
ctx = zg.Ctx(0)
s = zg.fr_from_int(0x5EED5EED)
g, gl = ctx.params_new(k, s)
bases = ctx.register_bases(gl)
gen = np.random.default_rng(1)


def uniform(m):
    a = gen.integers(0, 1 << 62, size=(m, 4), dtype=np.int64).astype(np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a


cases = {
    "uniform": uniform(n),
    "all_same_big": np.tile(uniform(1), (n, 1)),
    "ones": np.tile(zg.fr_from_int(1), (n, 1)),
    "zeros": np.zeros((n, 4), np.uint64),
    "bytes": np.stack([zg.fr_from_int(int(v)) for v in gen.integers(0, 256, size=256)])[gen.integers(0, 256, size=n)],
}
for name, sc in cases.items():
    batch = np.stack([sc] * 4)
    ctx.msm_batch(bases, batch)
    ctx.profile(True)
    for _ in range(3):
        ctx.msm_batch(bases, batch)
    st = ctx.profile_collect()
    ctx.profile(False)
    print(name, {kk: round(v[1] / v[0] * 1e3) for kk, v in sorted(st.items(), key=lambda kv: -kv[1][1])}, "us per launch (batch 4)")
