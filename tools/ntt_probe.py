"""Isolated NTT kernel timings (tuning aid; run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
import numpy as np, torch
import zg_halo2 as zg

k = int(sys.argv[1]) if len(sys.argv) > 1 else 14
n, ek = 1 << k, k + 3
en = 1 << ek
ctx = zg.Ctx(0)
gen = np.random.default_rng(1)
def rnd(shape):
    a = gen.integers(0, 1 << 62, size=shape + (4,), dtype=np.int64).astype(np.uint64); a[..., 3] &= np.uint64((1 << 60) - 1); return a
om, omi = zg.domain_omega(k)
div = zg.fr_from_int(pow(n, -1, zg.FR_MODULUS))
for batch in (1, 7, 21):
    a = torch.from_numpy(rnd((batch, n)).view(np.int64)).cuda()
    out = torch.empty((batch, en, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        ctx.ntt_batch_dev(a.data_ptr(), n, batch, k, omi, div)
        ctx.coeff_to_extended_batch_dev(a.data_ptr(), n, out.data_ptr(), en, batch, k, ek)
    ctx.sync()
    ctx.profile(True)
    for _ in range(5):
        ctx.ntt_batch_dev(a.data_ptr(), n, batch, k, omi, div)
    st1 = ctx.profile_collect()
    for _ in range(5):
        ctx.coeff_to_extended_batch_dev(a.data_ptr(), n, out.data_ptr(), en, batch, k, ek)
    st2 = ctx.profile_collect()
    ctx.profile(False)
    bf1 = batch * (n // 2) * k
    bf2 = batch * (en // 2) * ek
    t1 = sum(v[1] for v in st1.values()) / 5
    t2 = sum(v[1] for v in st2.values()) / 5
    print(f"batch {batch:2d}: iNTT 2^{k}: {t1*1e3:7.1f} us ({bf1/t1/1e6:6.1f} G butterflies/s) {dict((a,round(b[1]/5*1e3)) for a,b in st1.items())} | "
          f"coset 2^{ek}: {t2*1e3:7.1f} us ({bf2/t2/1e6:6.1f} G bf/s) {dict((a,round(b[1]/5*1e3)) for a,b in st2.items())}")
