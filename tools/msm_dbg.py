import sys, time, os
sys.path.insert(0,'0g-halo2_amd'); sys.path.insert(0,'oracle')
import numpy as np
import orc, zg_halo2 as zg
ctx = zg.Ctx(0)
ctx.set_msm_latency(False)
for k in (7, 8, 9, 10, 12):
    n = 1 << k
    prm = orc.params_new(k, 0xABCDEF)
    g = prm.g_lagrange_np()
    b = ctx.register_bases(g)
    for name, s in (("uniform", orc.fill_fr(3, n)), ("sparse", orc.fill_fr_sparse(4, n)), ("zeros", np.zeros((n,4),np.uint64))):
        t=time.time(); got = ctx.msm(b, s); dt=time.time()-t
        ok = np.array_equal(got, orc.msm(s, g))
        print(k, name, "c", b.window_bits, "ms %.2f" % (dt*1e3), ok, flush=True)
    b.free()
