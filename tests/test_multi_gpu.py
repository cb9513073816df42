"""N > 1 path: one MSM sharded by point range over ranks, partials all-gathered and added.

CPU (gloo, world_size 2): the per-rank partial comes from the oracle, the exchange + EC add are the
product's (multi_gpu.py + zg_g1_sum).  GPU: two gloo ranks each drive their own context on cuda:0."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world, port):
    for p in (os.path.join(ROOT, "0g-halo2_amd"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _cpu_worker(rank, world, port, n, ret):
    _init(rank, world, port)
    import multi_gpu
    import orc

    prm = orc.params_new(8)
    g = prm.g_np()[:n]
    ok = True
    for seed, sparse in ((1, False), (2, True)):
        s = (orc.fill_fr_sparse if sparse else orc.fill_fr)(seed, n)
        lo, hi = multi_gpu.shard_range(n, rank, world)
        got = multi_gpu.msm_sharded(lambda sh: orc.msm(sh, g[lo:hi]), s[lo:hi])
        ok = ok and np.array_equal(got, orc.msm(s, g))
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [256, 101, 1])
def test_sharded_msm_gloo_world2(n):
    world = 2
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_cpu_worker, args=(world, 29511 + n % 17, n, ret), nprocs=world, join=True)
        assert all(ret[r] for r in range(world))


def test_shard_range_partitions():
    sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
    import multi_gpu

    for n in (0, 1, 7, 16384, 131072 + 5):
        for world in (1, 2, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = multi_gpu.shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                cover += list(range(lo, hi)) if n < 100 else []
                if r:
                    assert lo == multi_gpu.shard_range(n, r - 1, world)[1]
            assert multi_gpu.shard_range(n, world - 1, world)[1] == n
            if n < 100:
                assert cover == list(range(n))


def _gpu_worker(rank, world, port, ret):
    _init(rank, world, port)
    import multi_gpu
    import orc
    import zg_halo2 as zg

    prm = orc.params_new(10)
    g = prm.g_np()
    n = g.shape[0]
    ctx = zg.Ctx(0)
    sb = multi_gpu.ShardedBases(ctx, g)
    ok = True
    for seed in (3, 4):
        s = orc.fill_fr_sparse(seed, n) if seed == 4 else orc.fill_fr(seed, n)
        ok = ok and np.array_equal(sb.msm(s), orc.msm(s, g, threads=4))
    ret[rank] = ok
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_msm_two_ranks_on_gpu():
    world = 2
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_gpu_worker, args=(world, 29533, ret), nprocs=world, join=True)
        assert all(ret[r] for r in range(world))
