"""N > 1 path: commitments sharded by point range over ranks, partial sums all-gathered and added.

CPU (gloo, world_size 2): the per-rank partial comes from the oracle, the exchange + EC additions are the
product's (multi_gpu.py: make_exchange / msm_sharded, zg_xyzz_sum_ranks / zg_g1_sum).  GPU: two gloo ranks each drive
their own context on cuda:0 -- a stand-alone sharded MSM, and create_proof with the shard INSIDE the prover
(zg_prover_set_shard): both ranks must end with the single-GPU proof bytes."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world, port):
    for p in (os.path.join(ROOT, "0g-halo2_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "harness"), os.path.join(ROOT, "tests")):
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


def _xyzz_of(orc, jac):
    """normalised Jacobian (uint64[12]) -> extended Jacobian (X, Y, ZZ, ZZZ) with a rank-dependent scaling, as the MSM
    kernels leave their results (Python integers)."""
    import zg_halo2 as zg

    q = zg.FQ_MODULUS
    x, y, z = (zg.fq_to_int(jac[4 * i:4 * i + 4]) for i in range(3))
    if z == 0:
        vals = (0, 1, 0, 0)
    else:
        lam = 0x1234567 + x % 97  # any non-zero scaling: (x l^2, y l^3, l^2, l^3)
        vals = (x * lam * lam % q, y * pow(lam, 3, q) % q, lam * lam % q, pow(lam, 3, q))
    return np.concatenate([zg.fq_from_int(v) for v in vals])


def _exchange_worker(rank, world, port, ret):
    _init(rank, world, port)
    import ctypes

    import multi_gpu
    import orc
    import zg_halo2 as zg

    zg.load()
    prm = orc.params_new(7)
    g = prm.g_np()
    n, count = g.shape[0], 3
    ex = multi_gpu.make_exchange(dist)
    lo, hi = multi_gpu.shard_range(n, rank, world)
    scal = [orc.fill_fr(40 + i, n) for i in range(count)]
    mine = np.stack([_xyzz_of(orc, orc.msm(s[lo:hi], g[lo:hi])) for s in scal])  # this rank's partial sums
    send = (ctypes.c_uint8 * mine.nbytes).from_buffer_copy(mine.tobytes())
    recv = (ctypes.c_uint8 * (mine.nbytes * world))()
    ex(send, recv)
    parts = np.frombuffer(bytes(recv), dtype=np.uint64).reshape(world, count, 16)
    assert np.array_equal(parts[rank], mine)
    got = zg.xyzz_sum_ranks(parts)
    ret[rank] = all(np.array_equal(got[i], orc.msm(scal[i], g)) for i in range(count))
    dist.barrier()
    dist.destroy_process_group()


def test_phase_exchange_and_rank_sum_gloo_world2():
    """What a sharded commitment phase does between the ranks: ONE all-gather of every commitment's partial sum
    (128-byte extended Jacobian points), then the per-commitment additions -- on the host, so it runs here."""
    world = 2
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_exchange_worker, args=(world, 29577, ret), nprocs=world, join=True)
        assert all(ret[r] for r in range(world))


def _sharded_prover_worker(rank, world, port, ret):
    _init(rank, world, port)
    import multi_gpu
    import orc
    import zg_halo2 as zg
    from circuits import toy_circuit

    ok = True
    ctx = zg.Ctx(0)
    for k, force_degree in ((7, 6), (9, None)):
        cs, asg, ilen = toy_circuit(k, force_degree=force_degree)
        img = cs.to_c()
        params = orc.params_new(k, 0xABCDEF)
        vk_repr = orc.fr_from_int(0x1234567)
        fixed, sigma = asg.fixed_values(), asg.sigma_values()
        pk = orc.ProvingKey(img, fixed, sigma, params, vk_repr)
        n = 1 << k
        lo, hi = multi_gpu.shard_range(n, rank, world)
        gb = ctx.register_bases(params.g_np()[lo:hi])
        glb = ctx.register_bases(params.g_lagrange_np()[lo:hi])
        prover = zg.Prover(ctx, img, fixed, sigma, gb, glb, vk_repr)
        adv, inst = asg.advice_values(), asg.instance_values(ilen)
        try:  # a slice of the SRS without a declared shard is refused
            prover.prove(adv, inst, 1)
            ok = False
        except zg.ZgError:
            pass
        prover.set_shard(rank, world, lo, multi_gpu.make_exchange(dist))
        prover.set_batch(3)
        for overlap in (True, False):
            prover.set_overlap(overlap)
            seeds = [21, 22, 23]
            got, _ = prover.prove_batch([adv] * 3, [inst] * 3, seeds)
            ok = ok and got == [orc.create_proof(pk, adv, inst, s)[1] for s in seeds]
        prover.close()
        gb.free()
        glb.free()
    ret[rank] = ok
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_commitments_inside_create_proof_two_ranks_on_gpu():
    world = 2
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_sharded_prover_worker, args=(world, 29541, ret), nprocs=world, join=True)
        assert all(ret[r] for r in range(world))


@pytest.mark.gpu
def test_rank_sums_on_the_device_match_the_host():
    """zg_xyzz_sum_ranks_dev (the additions behind the in-library all-gather) against zg_xyzz_sum_ranks, on partial sums
    of real shards: three 'ranks', five commitments, one of them the identity on every rank."""
    import ctypes

    sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import multi_gpu
    import orc
    import zg_halo2 as zg

    ctx = zg.Ctx(0)
    prm = orc.params_new(7)
    g = prm.g_np()
    n, world, count = g.shape[0], 3, 5
    scal = [orc.fill_fr(60 + i, n) for i in range(count - 1)] + [np.zeros((n, 4), np.uint64)]
    parts = np.zeros((world, count, 16), np.uint64)
    for r in range(world):
        lo, hi = multi_gpu.shard_range(n, r, world)
        for i, s in enumerate(scal):
            parts[r, i] = _xyzz_of(orc, orc.msm(s[lo:hi], g[lo:hi]))
    want = zg.xyzz_sum_ranks(parts)
    d_parts = torch.from_numpy(parts.view(np.int64).reshape(-1)).cuda()
    d_out = torch.zeros(count * 16, dtype=torch.int64, device="cuda")
    st = ctx.lib.zg_xyzz_sum_ranks_dev(ctx.h, ctypes.c_void_p(d_parts.data_ptr()), ctypes.c_size_t(world), ctypes.c_size_t(count),
                                       ctypes.c_void_p(d_out.data_ptr()))
    assert st == 0
    ctx.sync()
    summed = d_out.cpu().numpy().view(np.uint64).reshape(1, count, 16)
    assert np.array_equal(zg.xyzz_sum_ranks(summed), want)
    for i in range(count - 1):
        assert np.array_equal(want[i], orc.msm(scal[i], g))
    ctx.close()


@pytest.mark.gpu
def test_in_library_rccl_exchange_single_rank():
    """zg_prover_set_shard_rccl with a real RCCL communicator of ONE rank (two ranks on one GPU are refused by RCCL, and
    this pool has one GPU per box): librccl is bound at run time, every commitment phase goes through ncclAllGather on
    the prover's stream and the summing kernel, and the proofs are the oracle's -- in both scheduling forms, in a batch."""
    sys.path.insert(0, os.path.join(ROOT, "0g-halo2_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import multi_gpu
    import orc
    import zg_halo2 as zg
    from circuits import toy_circuit

    ctx = zg.Ctx(0)
    comm = multi_gpu.RcclComm(0, 1, 0)
    cs, asg, ilen = toy_circuit(8, force_degree=6)
    img = cs.to_c()
    params = orc.params_new(8, 0xABCDEF)
    vk_repr = orc.fr_from_int(0x1234567)
    fixed, sigma = asg.fixed_values(), asg.sigma_values()
    pk = orc.ProvingKey(img, fixed, sigma, params, vk_repr)
    prover = zg.Prover(ctx, img, fixed, sigma, params.g_np(), params.g_lagrange_np(), vk_repr)
    prover.set_shard_rccl(0, 1, 0, comm.handle)
    adv, inst = asg.advice_values(), asg.instance_values(ilen)
    prover.set_batch(3)  # (after the shard call: the gather buffer follows the slot count)
    for overlap in (True, False):
        prover.set_overlap(overlap)
        seeds = [71, 72, 73]
        got, _ = prover.prove_batch([adv] * 3, [inst] * 3, seeds)
        assert got == [orc.create_proof(pk, adv, inst, s)[1] for s in seeds]
    prover.close()
    comm.close()
    ctx.close()
