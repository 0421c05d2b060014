"""A CPU model of ONE camera stream sharded over G ranks (BASELINE configs[3]) -- TEST INFRASTRUCTURE, not product code.

What it states, with oracle-backed ranks and nothing but host arrays, is the algorithm both shard forms of the product rest on:
the stored model is a sequence of segments (one per fusing frame, in creation order), segment f lives on rank f mod G, a
surfel's global id is its position in that order, the index map is the MIN-reduction of the ranks' key maps (d24 << 32 | global id),
every rank fuses only the pixels whose winner it owns (the fused-pixel masks of the ranks are disjoint: their sum is their union),
and the owner of the frame's segment appends every candidate pixel nobody fused.  The product's form keeps the single-GPU SLOT
numbers as ids and runs the whole frame inside the HIP core (surfelmapping_amd/sharded.StreamShard, sm_shard_frame_device:
tests/test_shard_stream.py on the GPU); this model is what the CPU suite can run at world size 2 over gloo."""
from __future__ import annotations

import ctypes as C

import numpy as np

import oracle_lib as ol

KEY_EMPTY = np.uint64(0x7FFFFFFFFFFFFFFF)
NO_EXEMPT = 0xFFFFFFFF


class HostComm:
    """the three reductions on host arrays, over any object with allreduce_sum / allreduce_min / allgather
    (surfelmapping_amd.sharded.ThreadComm for thread ranks, TorchComm(device_index=None) over gloo for process ranks)"""

    def __init__(self, comm):
        self.c, self.rank, self.world = comm, comm.rank, comm.world

    def allreduce_sum(self, a):
        return self.c.allreduce_sum(a)

    def allgather(self, a):
        return self.c.allgather(a)

    def allreduce_min_keys(self, be):
        be.key_map_set(self.c.allreduce_min(be.key_map_get()))

    def allreduce_sum_mask(self, be):
        be.fused_mask_set(self.c.allreduce_sum(be.fused_mask_get().view(np.int64)).view(np.uint64))


class SegmentShardModel:
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



class OracleShardBackend:
    def __init__(self, cfg, rank, world):
        cfg.conflict_cap = 0            # like the HIP ranks: the cap is checked globally by SegmentShardModel
        self.o = ol.Oracle(cfg)
        self.L = ol.lib()
        for name, args in (("smo_begin_frame", [C.c_void_p] * 5), ("smo_end_frame", [C.c_void_p]),
                           ("smo_set_exempt_id", [C.c_void_p, C.c_int32]), ("smo_download_zbuf", [C.c_void_p] * 2),
                           ("smo_upload_index_ids", [C.c_void_p] * 3),
                           ("smo_download_data_pixels", [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
                           ("smo_filter_data", [C.c_void_p] * 2)):
            getattr(self.L, name).argtypes = args
            getattr(self.L, name).restype = C.c_int
        self.cfg, self.rank, self.world = cfg, rank, world
        self.P = cfg.width * cfg.height
        self.my_ticks = []
        self.pose = None

    @property
    def h(self):
        return self.o._h

    def begin_frame(self, rgb, depth, sem, pose):
        rgb = np.ascontiguousarray(rgb, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
        sem = np.ascontiguousarray(sem, np.uint8); self.pose = np.ascontiguousarray(pose, np.float32)
        rc = self.L.smo_begin_frame(self.h, rgb.ctypes.data, depth.ctypes.data, sem.ctypes.data, self.pose.ctypes.data)
        assert rc >= 0, rc
        return rc == 1

    def conflict(self, exempt_local, lstart_old):
        self.L.smo_set_exempt_id(self.h, -1 if exempt_local == 0xFFFFFFFF else int(exempt_local))
        c = self.cfg
        self.o.stage_process_conflict(self.pose, c.near_clip, c.far_clip, c.fuse_thresh, 0)
        n_conf = self.o.counts()["conflict_count"]
        self.o.stage_update_conflict(); self.o.stage_back_mapping(); self.o.stage_build_model_map()
        it = self.o.download_model()[:, 6]
        keep = np.array([int(np.sum(it == np.float32(t))) for t in self.my_ticks], np.uint32)
        assert keep.sum() == it.shape[0]
        return keep, n_conf

    def cull_splat(self, lstart_new, seg_gbase):
        c = self.cfg
        self.tick = self.o.counts()["tick"]
        self.o.stage_predict_indices(self.pose, self.tick, c.far_clip, c.time_delta)
        idx = self.o.download_index_map()[0].astype(np.int64)
        z = np.zeros(self.P, np.uint32)
        self.L.smo_download_zbuf(self.h, z.ctypes.data)
        has = z != 16777215
        ls = np.asarray(lstart_new, np.int64)
        if ls.shape[0] > 1:
            seg = np.clip(np.searchsorted(ls, idx, side="right") - 1, 0, ls.shape[0] - 2)
            gid = np.asarray(seg_gbase, np.int64)[seg] + (idx - ls[seg])
        else:
            gid = idx
        self.key = np.where(has, (z.astype(np.uint64) << np.uint64(32)) | gid.astype(np.uint64), KEY_EMPTY)
        self.lstart_new = ls

    def key_map_get(self):
        return self.key

    def key_map_set(self, a):
        self.key = np.ascontiguousarray(a, np.uint64)

    def associate(self, gbase):
        c, r, w = self.cfg, self.rank, self.world
        gb = np.asarray(gbase, np.int64)
        has = self.key != KEY_EMPTY
        gid = (self.key & np.uint64(0xFFFFFFFF)).astype(np.int64)
        F = gb.shape[0] - 1
        if F > 0:
            f = np.clip(np.searchsorted(gb, gid, side="right") - 1, 0, F - 1)
            mine = has & (gid > 0) & (f % w == r)
            local = np.where(mine, self.lstart_new[np.minimum(f // w, max(len(self.lstart_new) - 2, 0))] + gid - gb[f], 0)
        else:
            mine = np.zeros(self.P, bool)
            local = np.zeros(self.P, np.int64)
        idx_up = np.ascontiguousarray(local, np.int32)
        has_up = np.ascontiguousarray(mine, np.uint8)
        self.L.smo_upload_index_ids(self.h, idx_up.ctypes.data, has_up.ctypes.data)
        self.L.smo_set_exempt_id(self.h, -1)
        self.o.stage_data_associate(self.pose, self.tick, c.near_clip, c.far_clip)
        data = self.o.download_data()
        n = C.c_uint32()
        pix = np.zeros(max(data.shape[0], 1), np.int32)
        self.L.smo_download_data_pixels(self.h, pix.ctypes.data, pix.shape[0], C.byref(n))
        self.rec_pix = pix[:data.shape[0]].astype(np.int64)
        self.rec_local_fused = data[:, 5].view(np.int32) >= 0
        bits = np.zeros(((self.P + 63) // 64) * 64, np.uint8)
        bits[self.rec_pix[self.rec_local_fused]] = 1
        self.fmask = np.packbits(bits.reshape(-1, 64)[:, ::-1], axis=1).view(">u8").astype(np.uint64).reshape(-1)
        self.o.stage_update_fuse(); self.o.stage_back_mapping()

    def fused_mask_get(self):
        return self.fmask

    def fused_mask_set(self, a):
        self.fmask = np.ascontiguousarray(a, np.uint64)

    def append(self, here):
        words = self.fmask[self.rec_pix // 64]
        gfused = ((words >> (self.rec_pix % 64).astype(np.uint64)) & np.uint64(1)).astype(bool)
        U = int((~gfused).sum())
        Fz = int(sum(bin(int(x)).count("1") for x in self.fmask))
        keep = (~gfused & ~self.rec_local_fused) if here else np.zeros_like(gfused)
        keep = np.ascontiguousarray(keep, np.uint8)
        self.L.smo_filter_data(self.h, keep.ctypes.data)
        self.o.stage_concatenate(); self.o.stage_build_model_map()
        vis = self.o.counts()["visible_count"]
        self.L.smo_end_frame(self.h)
        if here:
            self.my_ticks.append(self.tick)
        return U, Fz, vis

    def download_model(self):
        return self.o.download_model()
