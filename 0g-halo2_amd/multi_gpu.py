"""Point-range sharding of the commitment MSMs across ranks (one process per GPU, torch.distributed).

Inside create_proof (zg_prover_set_shard): every rank holds the same witness and a slice of ParamsKZG::g /
::g_lagrange, multiplies only that slice, and per commitment phase ONE all-gather (make_exchange below) moves the
128-byte partial sums of all the phase's commitments; every rank adds them up (zg_xyzz_sum_ranks) and goes on with
the same transcript.  Stand-alone (ShardedBases, msm_sharded): the same for a single MSM.

Rank r registers bases[lo_r:hi_r) once (zg_bases_register) and, per MSM, multiplies its scalar slice;
the normalised 96-byte partial points are all-gathered (RCCL when the backend is "nccl", i.e. over
xGMI; gloo in CPU tests) and added on every rank with zg_g1_sum.  EC addition is not an ncclRedOp, so
this is a gather + local adds, never an all-reduce.  NTT, evaluate_h and the grand
products do not shard (north_star): multi-GPU proving otherwise means independent proofs per GPU.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

import zg_halo2 as zg


def shard_range(n: int, rank: int, world: int):
    """Contiguous point range of `rank`: sizes differ by at most one, empty shards allowed."""
    lo = rank * n // world
    hi = (rank + 1) * n // world
    return lo, hi


def gather_partials(part: np.ndarray, group=None, device=None) -> np.ndarray:
    """all_gather of one normalised Jacobian point (uint64[12]) -> uint64[world, 12]."""
    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(part, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return torch.stack(out).cpu().numpy().view(np.uint64)


def msm_sharded(local_msm, scalars_shard: np.ndarray, group=None, device=None) -> np.ndarray:
    """local_msm(scalars_shard) -> this rank's partial (normalised Jacobian); returns the full MSM."""
    part = local_msm(scalars_shard)
    parts = gather_partials(part, group, device)
    return zg.g1_sum(parts)


class ShardedBases:
    """This rank's slice of a base set, resident on its GPU."""

    def __init__(self, ctx: zg.Ctx, bases: np.ndarray, group=None, window_bits: int = 0):
        self.ctx, self.group = ctx, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n = bases.shape[0]
        self.lo, self.hi = shard_range(self.n, self.rank, self.world)
        self.local = ctx.register_bases(bases[self.lo:self.hi], window_bits) if self.hi > self.lo else None

    def msm(self, scalars: np.ndarray, device=None) -> np.ndarray:
        assert scalars.shape[0] == self.n

        def local(s):
            if self.local is None:
                return zg.g1_sum(np.zeros((0, 12), np.uint64))  # identity
            return self.ctx.msm(self.local, s)

        return msm_sharded(local, scalars[self.lo:self.hi], self.group, device)


def make_exchange(dist_mod=dist, device=None, group=None):
    """The all-gather zg_prover_set_shard asks for: exchange(send, recv) with `send` this rank's bytes and `recv`
    room for world x len(send), rank order.  device = a cuda device: the bytes travel through HBM and RCCL (xGMI
    between the GPUs of a node); None: gloo on the host."""
    def exchange(send, recv):
        world = dist_mod.get_world_size(group)
        t = torch.frombuffer(send, dtype=torch.uint8).clone()
        if device is not None:
            t = t.to(device)
            out = torch.empty(world * t.numel(), dtype=torch.uint8, device=device)
            dist_mod.all_gather_into_tensor(out, t, group=group)
            out = out.cpu()
        else:
            parts = [torch.empty_like(t) for _ in range(world)]
            dist_mod.all_gather(parts, t, group=group)
            out = torch.cat(parts)
        torch.frombuffer(recv, dtype=torch.uint8).copy_(out)

    return exchange


class RcclComm:
    """A raw RCCL communicator for zg_prover_set_shard_rccl (torch.distributed keeps its own to itself): rank 0 draws
    the unique id, `dist_mod` (any backend) carries its 128 bytes to the others, every rank joins with ncclCommInitRank
    on its own device.  world = 1 needs no dist_mod.  `handle` is the ncclComm_t."""

    def __init__(self, rank: int, world: int, device_index: int, dist_mod=None, group=None):
        import ctypes

        class UniqueId(ctypes.Structure):  # ncclUniqueId: 128 opaque bytes (c_ubyte: a c_char array would read as a
            _fields_ = [("internal", ctypes.c_ubyte * 128)]  # NUL-terminated string and truncate the id)

        self._rccl = ctypes.CDLL("librccl.so.1")
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipSetDevice(ctypes.c_int(device_index)) == 0
        uid = UniqueId()
        if rank == 0:
            assert self._rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0, "ncclGetUniqueId failed"
        if world > 1:
            box = [ctypes.string_at(ctypes.byref(uid), 128) if rank == 0 else None]
            dist_mod.broadcast_object_list(box, src=0, group=group)
            ctypes.memmove(ctypes.byref(uid), box[0], 128)
        comm = ctypes.c_void_p()
        self._rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
        st = self._rccl.ncclCommInitRank(ctypes.byref(comm), ctypes.c_int(world), uid, ctypes.c_int(rank))
        assert st == 0, f"ncclCommInitRank failed with {st}"
        self.handle = comm.value
        self.rank, self.world = rank, world

    def count(self) -> int:
        """the communicator's size as RCCL reports it (ncclCommCount)"""
        import ctypes

        n = ctypes.c_int(0)
        self._rccl.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        assert self._rccl.ncclCommCount(ctypes.c_void_p(self.handle), ctypes.byref(n)) == 0
        return n.value

    def close(self):
        if getattr(self, "handle", None):
            import ctypes

            self._rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            self._rccl.ncclCommDestroy(ctypes.c_void_p(self.handle))
            self.handle = None
