"""Oracle-backed rank for the sharded-mode tests (CPU): implements the per-rank stage interface of
surfelmapping_amd/sharded.py with the CPU oracle's pass functions plus the small shard helpers of
oracle/smo.h.  TEST INFRASTRUCTURE ONLY (the product's ranks are HipShardBackend)."""
import ctypes as C

import numpy as np

import oracle_lib as ol

KEY_EMPTY = np.uint64(0x7FFFFFFFFFFFFFFF)


class OracleShardBackend:
    def __init__(self, cfg, rank, world):
        cfg.conflict_cap = 0            # like the HIP ranks: the cap is checked globally by ShardedMapper
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
