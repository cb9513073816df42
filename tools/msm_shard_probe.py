"""What point-range sharding of ONE MSM can buy: device time of a single-vector MSM of n
points against the time of its n/G shard on one GPU (the per-GPU leg of a G-way split; the 96-byte
all_gather and G-1 EC adds come on top).  Run on the GPU box:  python tools/msm_shard_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
import numpy as np
import torch
import zg_halo2 as zg

ctx = zg.Ctx(0)
gen = np.random.default_rng(1)
for k in (14, 15, 17):
    n = 1 << k
    _, gl = ctx.params_new(k, zg.fr_from_int(0x5EED5EED))
    sc = gen.integers(0, 1 << 62, size=(n, 4), dtype=np.int64).astype(np.uint64)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    d_s = torch.from_numpy(sc.view(np.int64)).cuda()
    d_out = torch.empty(64, dtype=torch.int64, device="cuda")
    row = []
    for G in (1, 2, 4, 8):
        m = n // G
        bases = ctx.register_bases(gl[:m])
        for _ in range(3):
            ctx.msm_batch_dev(bases, d_s.data_ptr(), n, 1, m, d_out.data_ptr())
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.msm_batch_dev(bases, d_s.data_ptr(), n, 1, m, d_out.data_ptr())
        ctx.sync()
        row.append((G, (time.perf_counter() - t0) / 20 * 1e6))
        bases.free()
    base = row[0][1]
    print(f"k={k}: " + "  ".join(f"{G}-way shard {us:7.1f} us (x{base / us:4.2f})" for G, us in row), flush=True)
