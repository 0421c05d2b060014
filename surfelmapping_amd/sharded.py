"""ONE camera stream sharded over G GPUs, bit-identical to the single-GPU result
(BASELINE configs[3]; DESIGN.md "Multi-GPU").

Partitioning.  The stored model is always sorted by creation frame (survivors keep their order,
new surfels are appended: src/GlobalModel.cpp:517-637), so it is a sequence of *segments*, one
per fusing frame.  Segment f lives on rank f % G.  The concatenation of all segments in frame
order is exactly the single-GPU model, hence a surfel's GLOBAL id = (survivors of all earlier
segments) + its rank inside its segment -- computable from G small integers per frame.  No
surfel ever moves between GPUs.

Per frame every rank runs the same five stages on its slice; between them three reductions
cross the ranks (RCCL over xGMI on device buffers, or any stand-in with the same semantics):

    conflict      p2/p3 on the slice            -> survivors per segment, conflict count
       all-reduce(sum)  [segment survivor counts | conflicts]     (F+1 integers)
    cull_splat    p4..p6 under global ids        -> local 64-bit key map  d24<<32 | global id
       all-reduce(min)  key map                                    (W*H x 8 B)
    associate     p8..p10 for the pixels whose winner this rank owns -> fused-pixel ballot words
       all-reduce(sum)  fused mask (bit sets are disjoint: sum == union)   (W*H/8 B)
    append        p11 on rank f % G only (it derives every new surfel from the replicated frame)

The W*H conflict cap (src/GlobalModel.cpp:54-57) depends on the global conflict order and is not
evaluated per shard: a frame whose global conflict count exceeds W*H raises instead of returning
a result that could differ from the reference.
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

    # device buffers: staged through the host
    def allreduce_min_keys(self, be):
        be.key_map_set(self.allreduce_min(be.key_map_get()))

    def allreduce_sum_mask(self, be):
        be.fused_mask_set(self.allreduce_sum(be.fused_mask_get()))


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

    def _alias(self, ptr, n):
        import torch
        from .dist import _DevArray
        return torch.as_tensor(_DevArray(ptr, (n,), "<i8"), device=self._dev())

    def allreduce_min_keys(self, be):
        if self.device_index is not None and hasattr(be, "key_map_device"):
            import torch
            ptr, n = be.key_map_device()
            self.dist.all_reduce(self._alias(ptr, n), op=self.dist.ReduceOp.MIN, group=self.group)
            torch.cuda.synchronize(self._dev())
        else:
            be.key_map_set(self.allreduce_min(be.key_map_get()))

    def allreduce_sum_mask(self, be):
        if self.device_index is not None and hasattr(be, "fused_mask_device"):
            import torch
            ptr, n = be.fused_mask_device()
            self.dist.all_reduce(self._alias(ptr, n), op=self.dist.ReduceOp.SUM, group=self.group)
            torch.cuda.synchronize(self._dev())
        else:
            be.fused_mask_set(self.allreduce_sum(be.fused_mask_get().view(np.int64)).view(np.uint64))


# ---------------------------------------------------------------------------------------------
# per-rank compute on the HIP core
# ---------------------------------------------------------------------------------------------
class HipShardBackend:
    def __init__(self, sm, rank: int, world: int):
        self.sm = sm
        self.P = sm.P
        sm.shard_configure(rank, world)

    def begin_frame(self, rgb, depth, sem, pose) -> bool:
        return self.sm.shard_begin_frame(rgb, depth, sem, pose) == 1

    def conflict(self, exempt_local, lstart_old):
        return self.sm.shard_conflict(exempt_local, lstart_old)

    def cull_splat(self, lstart_new, seg_gbase):
        self.sm.shard_cull_splat(lstart_new, seg_gbase)

    def associate(self, gbase):
        self.sm.shard_associate(gbase)

    def append(self, here: bool):
        self.sm.shard_append(here)
        c = self.sm.counts()
        return c["unstable_count"], c["fused_count"], c["visible_count"]

    def key_map_device(self):
        return self.sm.key_map_device_ptr(), self.P

    def fused_mask_device(self):
        return self.sm.fused_mask_device_ptr()

    def key_map_get(self):
        return self.sm.device_download(self.sm.key_map_device_ptr(), self.P * 8, np.uint64)

    def key_map_set(self, a):
        self.sm.device_upload(self.sm.key_map_device_ptr(), np.ascontiguousarray(a, np.uint64))

    def fused_mask_get(self):
        p, n = self.sm.fused_mask_device_ptr()
        return self.sm.device_download(p, n * 8, np.uint64)

    def fused_mask_set(self, a):
        p, _ = self.sm.fused_mask_device_ptr()
        self.sm.device_upload(p, np.ascontiguousarray(a, np.uint64))

    def download_model(self):
        return self.sm.download_model()


# ---------------------------------------------------------------------------------------------
# the SPMD frame loop
# ---------------------------------------------------------------------------------------------
class ShardedMapper:
    """Every rank constructs one with its own backend + communicator and calls process_frame with
    the same arguments (the frame is replicated: <= 2.8 MB at KITTI size)."""

    def __init__(self, backend, comm, n_pixels: int, conflict_cap: bool = True, collect_stats: bool = True):
        self.be, self.comm = backend, comm
        self.P = n_pixels
        self.conflict_cap = conflict_cap
        self.collect_stats = collect_stats
        self.cnt: list[int] = []          # survivors of every global segment (identical on all ranks)
        self.tick = 0
        self.last = {}

    def _mine(self, F):
        return [f for f in range(F) if f % self.comm.world == self.comm.rank]

    def process_frame(self, rgb, depth, sem, pose):
        r, w = self.comm.rank, self.comm.world
        go = self.be.begin_frame(rgb, depth, sem, pose)
        self.tick += 1
        if not go:
            self.last = dict(count=sum(self.cnt), offset=sum(self.cnt), conflict_count=0, unstable_count=0,
                             fused_count=0, data_count=0, visible_count=0, tick=self.tick)
            return self.last
        F = len(self.cnt)
        mine = self._mine(F)
        lstart_old = np.concatenate([[0], np.cumsum([self.cnt[f] for f in mine])]).astype(np.uint32)
        f0 = next((f for f in range(F) if self.cnt[f] > 0), None)      # global id 0 = first surfel of that segment
        exempt = int(lstart_old[f0 // w]) if (f0 is not None and f0 % w == r) else NO_EXEMPT
        seg_keep, c_local = self.be.conflict(exempt, lstart_old)
        vec = np.zeros(F + 1, np.int64)
        vec[mine] = seg_keep
        vec[F] = c_local
        vec = self.comm.allreduce_sum(vec)
        cnt_new = vec[:F]
        c_total = int(vec[F])
        if self.conflict_cap and c_total > self.P:
            raise RuntimeError(f"{c_total} conflicts > W*H = {self.P}: the reference's conflict cap would truncate them in "
                               "global surfel order, which a sharded cull cannot reproduce")
        gbase = np.concatenate([[0], np.cumsum(cnt_new)]).astype(np.uint32)          # F + 1
        lstart_new = np.concatenate([[0], np.cumsum(cnt_new[mine])]).astype(np.uint32)
        self.be.cull_splat(lstart_new, gbase[mine].astype(np.uint32))
        self.comm.allreduce_min_keys(self.be)
        self.be.associate(gbase)
        self.comm.allreduce_sum_mask(self.be)
        U, Fz, vis = self.be.append(F % w == r)
        self.cnt = [int(x) for x in cnt_new] + [int(U)]
        if self.collect_stats:
            vis = int(self.comm.allreduce_sum(np.array([vis], np.int64))[0])
        offset = int(cnt_new.sum())
        self.last = dict(count=offset + int(U), offset=offset, conflict_count=c_total, unstable_count=int(U),
                         fused_count=int(Fz), data_count=int(U) + int(Fz), visible_count=vis, tick=self.tick)
        return self.last

    def counts(self):
        return dict(self.last)

    def gather_global_model(self) -> np.ndarray:
        """All segments in frame order = the single-GPU model (AoS float32 [count][12])."""
        w = self.comm.world
        locals_ = self.comm.allgather(np.ascontiguousarray(self.be.download_model(), np.float32))
        F = len(self.cnt)
        parts = []
        cursor = [0] * w
        for f in range(F):
            owner, n = f % w, self.cnt[f]
            parts.append(locals_[owner][cursor[owner]:cursor[owner] + n])
            cursor[owner] += n
        for rk in range(w):
            assert cursor[rk] == locals_[rk].shape[0], "segment table out of step with the local model"
        return np.concatenate(parts, axis=0) if parts else np.zeros((0, 12), np.float32)


# ---------------------------------------------------------------------------------------------
# in-stream form: slot-addressed sharding, the whole frame inside the HIP core (sm_shard_frame*)
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
