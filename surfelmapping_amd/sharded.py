"""ONE camera stream sharded over G GPUs, bit-identical to the single-GPU result (BASELINE configs[3]; SURVEY.md 8e;
DESIGN.md 6), and the small communicators the multi-rank tests and the rig (surfelmapping_amd/dist.py) share.

Partitioning.  The stored model is always sorted by creation frame (survivors keep their order, new surfels are appended:
src/GlobalModel.cpp:517-637), so it is a sequence of *segments*, one per fusing frame.  Segment f lives on rank f % G, at
the slot numbers the single-GPU run would use (slot = offset + candidate pixels before the pixel: a function of the
replicated frame), so no surfel ever moves between GPUs and no id needs translating.  The whole frame runs inside the HIP
core (sm_shard_frame_device): per frame an all-reduce(min) of the W*H x 8 B key map, an all-reduce(sum) of the fused-pixel
mask + 3 counters, and -- once the model has more slots than pixels, with the conflict cap on -- an all-reduce(sum) of the
conflict masks, so that the first W*H conflicts in the surfel order of ALL ranks take effect (src/GlobalModel.cpp:54-57).
"""
from __future__ import annotations

import threading

import numpy as np

KEY_EMPTY = np.uint64(0x7FFFFFFFFFFFFFFF)
NO_EXEMPT = 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------
# communicators
# ---------------------------------------------------------------------------------------------
class ThreadGroup:
    def __init__(self, world: int):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class ThreadComm:
    """In-process stand-in for the collectives: G ranks = G threads (each may drive its own HIP
    context on the same GPU).  Used by the tests and for rehearsing on a 1-GPU box."""

    def __init__(self, group: ThreadGroup, rank: int):
        self.g, self.rank, self.world = group, rank, group.world

    def _exchange(self, value):
        self.g.slots[self.rank] = value
        self.g.barrier.wait()
        vals = list(self.g.slots)
        self.g.barrier.wait()
        return vals

    def allreduce_sum(self, a: np.ndarray) -> np.ndarray:
        return np.sum(np.stack(self._exchange(a)), axis=0).astype(a.dtype)

    def allreduce_min(self, a: np.ndarray) -> np.ndarray:
        return np.minimum.reduce(np.stack(self._exchange(a)))

    def allgather(self, a):
        return self._exchange(a)


class TorchComm:
    """torch.distributed collectives: 'gloo' on host arrays (CPU tests) or 'nccl' (= RCCL) directly
    on the HIP core's device buffers, aliased as tensors without a copy."""

    def __init__(self, device_index=None, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device_index = device_index            # None -> host path

    def _dev(self):
        import torch
        return torch.device("cuda", self.device_index)
    def allreduce_sum(self, a: np.ndarray) -> np.ndarray:
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a).astype(np.int64))
        if self.device_index is not None:
            t = t.to(self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy().astype(a.dtype)

    def allreduce_min(self, a: np.ndarray) -> np.ndarray:
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy())   # keys < 2^63: order preserved
        if self.device_index is not None:
            t = t.to(self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return t.cpu().numpy().view(a.dtype)

    def allgather(self, a):
        out = [None] * self.world
        self.dist.all_gather_object(out, a, group=self.group)
        return out


# ---------------------------------------------------------------------------------------------
# slot-addressed sharding, the whole frame inside the HIP core (sm_shard_frame*)
# ---------------------------------------------------------------------------------------------
class ThreadCollective:
    """The collective of sm_shard_set_collective for G contexts driven by G threads of one process (the ranks may share one
    GPU): staged through the host, for tests and for rehearsing the multi-rank path on a 1-GPU box.  Production uses
    RCCL on the context's stream (SurfelMap.shard_rccl_init)."""

    def __init__(self, group: ThreadGroup, rank: int, sm):
        self.comm = ThreadComm(group, rank)
        self.sm = sm

    def __call__(self, send, recv, count, op):
        from .capi import SM_COLL_GATHER, SM_COLL_MIN
        a = self.sm.device_download(send, count * 8, np.uint64)          # waits for the context's stream
        vals = self.comm._exchange(a)
        if op == SM_COLL_GATHER:
            r = np.concatenate(vals)                                      # rank q's words at recv + q * count
        elif op == SM_COLL_MIN:
            r = np.minimum.reduce(np.stack(vals))
        else:
            r = np.sum(np.stack(vals), axis=0, dtype=np.uint64)
        self.sm.device_upload(recv, np.ascontiguousarray(r, np.uint64))
        return 0


class TorchCollective:
    """The collective of sm_shard_set_collective for ranks that are PROCESSES joined by a torch.distributed group: staged through the
    host like ThreadCollective, the reduction itself by torch (`device` None: host tensors, i.e. a gloo group; a torch device:
    tensors there, i.e. torch's own NCCL / RCCL group).  For rehearsing the multi-process path where the core's RCCL binding cannot
    run -- several ranks on one GPU (`bench.py --gpus N --rehearse-one-gpu`, tests/test_dist_gpu.py) -- and as bench.py's fallback
    should that binding fail to initialise; production uses RCCL on the context's stream (SurfelMap.shard_rccl_init)."""

    def __init__(self, sm, group=None, device=None):
        import torch.distributed as dist
        self.sm, self.dist, self.group, self.device = sm, dist, group, device
        self.world = dist.get_world_size(group)

    def __call__(self, send, recv, count, op):
        import torch
        from .capi import SM_COLL_GATHER, SM_COLL_MIN
        a = self.sm.device_download(send, count * 8, np.uint64)          # waits for the context's stream
        t = torch.from_numpy(a.view(np.int64).copy())                     # (keys < 2^63: order kept; sums wrap like u64)
        if self.device is not None:
            t = t.to(self.device)
        if op == SM_COLL_GATHER:
            parts = [torch.empty_like(t) for _ in range(self.world)]
            self.dist.all_gather(parts, t, group=self.group)
            t = torch.cat(parts)                                          # rank q's words at recv + q * count
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN if op == SM_COLL_MIN else self.dist.ReduceOp.SUM, group=self.group)
        self.sm.device_upload(recv, np.ascontiguousarray(t.cpu().numpy().view(np.uint64)))
        return 0


GlooCollective = TorchCollective      # (host tensors: the name the rehearsal uses)


class StreamShard:
    """One rank of ONE camera stream sharded over `world` GPUs, in-stream form: every rank calls
    process_frame with the same arguments; counters are identical on all ranks after every frame.

    collective: None with world == 1 (identity), a callable (send_ptr, recv_ptr, count_u64, op) -> 0, or the string
    "rccl" together with `rccl_id` (capi.rccl_unique_id() of one rank, handed to all)."""

    def __init__(self, sm, rank: int, world: int, collective=None, rccl_id: bytes | None = None):
        self.sm, self.rank, self.world = sm, rank, world
        sm.shard_stream_configure(rank, world)
        if collective == "rccl":
            sm.shard_rccl_init(rccl_id)
        elif collective is not None:
            sm.shard_set_collective(collective)

    def process_frame(self, rgb, depth, sem, pose):
        self.sm.shard_frame(rgb, depth, sem, pose)
        return self.sm.counts()

    def counts(self):
        return self.sm.counts()

    def export_dense(self) -> np.ndarray:
        """this rank's surfels at their positions in the union, zeros elsewhere (collective: it compacts)"""
        return self.sm.shard_export_dense()

    @staticmethod
    def union(planes) -> np.ndarray:
        """the single GlobalModel from the dense planes of all ranks (their supports are disjoint: integer sum == union)"""
        acc = np.zeros(planes[0].shape, np.uint32)
        for p in planes:
            acc += np.ascontiguousarray(p, np.float32).view(np.uint32)
        return acc.view(np.float32)
