"""ctypes binding of the CPU oracle (oracle/libsmo.so).  TEST INFRASTRUCTURE ONLY.

The oracle is the checker for the HIP path; nothing in surfelmapping_amd/ imports this.
bench.py's cpu_baseline leg and __graft_entry__.smoke() are the only other users.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libsmo.so")


class SmoConfig(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("near_clip", C.c_float), ("far_clip", C.c_float), ("fuse_thresh", C.c_float),
        ("max_sqrt_vertices", C.c_int32), ("time_delta", C.c_int32),
        ("stereo_border", C.c_float), ("preprocess", C.c_int32), ("conflict_cap", C.c_int32),
    ]


class SmoCounts(C.Structure):
    _fields_ = [
        ("count", C.c_uint32), ("offset", C.c_uint32), ("data_count", C.c_uint32),
        ("conflict_count", C.c_uint32), ("unstable_count", C.c_uint32),
        ("fused_count", C.c_uint32), ("visible_count", C.c_uint32), ("tick", C.c_int32),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build(force: bool = False) -> str:
    src = [os.path.join(ORACLE_DIR, f) for f in ("smo.c", "smo.h", "Makefile")]
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "libsmo.so"])
    return LIB_PATH


OMP_LIB_PATH = os.path.join(ORACLE_DIR, "libsmo_omp.so")     # all-core build of the same source (bit-identical results)
_libs = {}


def lib(path=None):
    """The oracle library (default: the serial contract build); `path` selects another build of oracle/smo.c."""
    path = path or LIB_PATH
    if path not in _libs:
        build()
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", os.path.basename(path)])
        L = C.CDLL(path)
        fp = C.POINTER(C.c_float)
        L.smo_create.restype = C.c_void_p
        L.smo_create.argtypes = [C.POINTER(SmoConfig)]
        L.smo_destroy.argtypes = [C.c_void_p]
        L.smo_default_config.argtypes = [C.POINTER(SmoConfig), C.c_int, C.c_int] + [C.c_float] * 4
        for name in ("smo_process_frame", "smo_clean_points", "smo_reset", "smo_get_counts",
                     "smo_download_model", "smo_upload_model", "smo_download_index_map",
                     "smo_download_depth", "smo_download_data", "smo_set_frame", "smo_set_tick",
                     "smo_stage_process_conflict", "smo_stage_update_conflict",
                     "smo_stage_back_mapping", "smo_stage_build_model_map",
                     "smo_stage_predict_indices", "smo_stage_data_associate",
                     "smo_stage_update_fuse", "smo_stage_concatenate"):
            getattr(L, name).restype = C.c_int
        L.smo_process_frame.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.smo_clean_points.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        L.smo_reset.argtypes = [C.c_void_p]
        L.smo_get_counts.argtypes = [C.c_void_p, C.POINTER(SmoCounts)]
        L.smo_download_model.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.smo_upload_model.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.smo_download_index_map.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.smo_download_depth.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.smo_download_data.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.smo_set_frame.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        L.smo_set_tick.argtypes = [C.c_void_p, C.c_int32]
        L.smo_stage_process_conflict.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                                 C.c_float, C.c_int]
        for name in ("smo_stage_update_conflict", "smo_stage_back_mapping",
                     "smo_stage_build_model_map", "smo_stage_update_fuse", "smo_stage_concatenate"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.smo_stage_predict_indices.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int]
        L.smo_stage_data_associate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float]
        L.smo_metricise.argtypes = [C.POINTER(SmoConfig), C.c_void_p, C.c_void_p]
        L.smo_filter_depth.argtypes = [C.POINTER(SmoConfig), C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.smo_smooth_depth.argtypes = [C.POINTER(SmoConfig), C.c_void_p, C.c_void_p, C.c_void_p]
        L.smo_remove_movings.argtypes = [C.POINTER(SmoConfig)] + [C.c_void_p] * 5
        L.smo_encode_color.restype = C.c_float
        L.smo_encode_color.argtypes = [C.c_float] * 3 + [C.c_uint32]
        L.smo_get_radius.restype = C.c_float
        L.smo_get_radius.argtypes = [C.c_float] * 4
        L.smo_acosf.restype = C.c_float
        L.smo_acosf.argtypes = [C.c_float]
        L.smo_expf.restype = C.c_float
        L.smo_expf.argtypes = [C.c_float]
        L.smo_render_image.restype = C.c_int
        L.smo_render_image.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_void_p] * 2
        L.smo_raw_cloud.restype = C.c_int
        L.smo_raw_cloud.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.smo_invert4.argtypes = [fp, fp]
        L.smo_mul4.argtypes = [fp, fp, fp]
        _libs[path] = L
    return _libs[path]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed rc={rc}")


def make_config(width, height, fx, fy, cx, cy, **over) -> SmoConfig:
    c = SmoConfig()
    lib().smo_default_config(C.byref(c), width, height, fx, fy, cx, cy)
    for k, v in over.items():
        if not hasattr(c, k):
            raise KeyError(k)
        setattr(c, k, v)
    return c


class Oracle:
    """Thin object wrapper with the same method names as the product binding."""

    def __init__(self, cfg: SmoConfig, libpath=None):
        self.cfg = cfg
        self.W, self.H = cfg.width, cfg.height
        self.P = self.W * self.H
        self._L = lib(libpath)
        self._h = self._L.smo_create(C.byref(cfg))
        if not self._h:
            raise RuntimeError("smo_create failed")

    def close(self):
        if self._h:
            self._L.smo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- frame level
    def process_frame(self, rgb, depth, sem, pose, allow=(0,)):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        sem = np.ascontiguousarray(sem, np.uint8)
        pose = np.ascontiguousarray(pose, np.float32)
        rc = self._L.smo_process_frame(self._h, _ptr(rgb), _ptr(depth), _ptr(sem), _ptr(pose))
        if rc not in allow:
            _chk(rc, "process_frame")
        return rc

    def clean_points(self, depth, sem, pose):
        depth = np.ascontiguousarray(depth, np.uint16)
        sem = np.ascontiguousarray(sem, np.uint8)
        pose = np.ascontiguousarray(pose, np.float32)
        _chk(self._L.smo_clean_points(self._h, _ptr(depth), _ptr(sem), _ptr(pose)), "clean_points")

    def reset(self):
        _chk(self._L.smo_reset(self._h), "reset")

    STAGES = ("preprocess", "processConflict", "updateConflict", "backMapping", "buildModelMap", "predictIndices", "dataAssociate",
              "updateFuse", "concatenate")

    def stage_seconds(self, reset=False) -> dict:
        """wall seconds process_frame spent per pass since the last reset"""
        out = (C.c_double * 9)()
        self._L.smo_stage_seconds.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        _chk(self._L.smo_stage_seconds(self._h, out, 1 if reset else 0), "stage_seconds")
        return dict(zip(self.STAGES, [float(x) for x in out]))

    def counts(self) -> dict:
        c = SmoCounts()
        _chk(self._L.smo_get_counts(self._h, C.byref(c)), "get_counts")
        return c.as_dict()

    def download_model(self) -> np.ndarray:
        n = C.c_uint32()
        _chk(self._L.smo_download_model(self._h, None, 0, C.byref(n)), "download_model")
        out = np.zeros((n.value, 12), np.float32)
        _chk(self._L.smo_download_model(self._h, _ptr(out), n.value, C.byref(n)), "download_model")
        return out

    def upload_model(self, m):
        m = np.ascontiguousarray(m, np.float32)
        _chk(self._L.smo_upload_model(self._h, _ptr(m), m.shape[0]), "upload_model")

    def download_index_map(self):
        P = self.P
        idx = np.zeros(P, np.int32)
        vc = np.zeros((P, 4), np.float32)
        ct = np.zeros((P, 4), np.float32)
        nr = np.zeros((P, 4), np.float32)
        _chk(self._L.smo_download_index_map(self._h, _ptr(idx), _ptr(vc), _ptr(ct), _ptr(nr)), "index_map")
        return idx, vc, ct, nr

    def download_raw_cloud(self) -> np.ndarray:
        """FeedbackBuffer "RAW" of the last processed frame (time stamp = its tick)."""
        t = self.counts()["tick"] - 1
        n = C.c_uint32()
        _chk(self._L.smo_raw_cloud(self._h, t, None, 0, C.byref(n)), "raw_cloud")
        out = np.zeros((n.value, 12), np.float32)
        if n.value:
            _chk(self._L.smo_raw_cloud(self._h, t, _ptr(out), n.value, C.byref(n)), "raw_cloud")
        return out

    def download_depth(self, which=0):
        out = np.zeros((self.H, self.W), np.float32)
        _chk(self._L.smo_download_depth(self._h, which, _ptr(out)), "download_depth")
        return out

    def download_data(self):
        n = C.c_uint32()
        _chk(self._L.smo_download_data(self._h, None, 0, C.byref(n)), "download_data")
        out = np.zeros((n.value, 12), np.float32)
        _chk(self._L.smo_download_data(self._h, _ptr(out), n.value, C.byref(n)), "download_data")
        return out

    def render_image(self, view, w, h, fx, fy, cx, cy):
        view = np.ascontiguousarray(view, np.float32)
        bgr = np.zeros((h, w, 3), np.uint8)
        sem = np.zeros((h, w), np.uint8)
        _chk(self._L.smo_render_image(self._h, _ptr(view), w, h, fx, fy, cx, cy, _ptr(bgr), _ptr(sem)), "render_image")
        return bgr, sem

    # -- stage level
    def set_frame(self, rgb=None, depth_metric=None, sem=None):
        rgb = None if rgb is None else np.ascontiguousarray(rgb, np.uint8)
        dm = None if depth_metric is None else np.ascontiguousarray(depth_metric, np.float32)
        sem = None if sem is None else np.ascontiguousarray(sem, np.uint8)
        _chk(self._L.smo_set_frame(self._h, _ptr(rgb), _ptr(dm), _ptr(sem)), "set_frame")

    def set_tick(self, tick):
        _chk(self._L.smo_set_tick(self._h, tick), "set_tick")

    def stage_process_conflict(self, pose, min_depth, max_depth, fuse_thresh=0.0, is_clean=0):
        pose = np.ascontiguousarray(pose, np.float32)
        _chk(self._L.smo_stage_process_conflict(self._h, _ptr(pose), min_depth, max_depth,
                                              fuse_thresh, is_clean), "process_conflict")

    def stage_update_conflict(self):
        _chk(self._L.smo_stage_update_conflict(self._h), "update_conflict")

    def stage_back_mapping(self):
        _chk(self._L.smo_stage_back_mapping(self._h), "back_mapping")

    def stage_build_model_map(self):
        _chk(self._L.smo_stage_build_model_map(self._h), "build_model_map")

    def stage_predict_indices(self, pose, time, depth_cutoff, time_delta):
        pose = np.ascontiguousarray(pose, np.float32)
        _chk(self._L.smo_stage_predict_indices(self._h, _ptr(pose), time, depth_cutoff, time_delta),
             "predict_indices")

    def stage_data_associate(self, pose, time, dmin, dmax):
        pose = np.ascontiguousarray(pose, np.float32)
        _chk(self._L.smo_stage_data_associate(self._h, _ptr(pose), time, dmin, dmax), "data_associate")

    def stage_update_fuse(self):
        _chk(self._L.smo_stage_update_fuse(self._h), "update_fuse")

    def stage_concatenate(self, allow=(0,)):
        rc = self._L.smo_stage_concatenate(self._h)
        if rc not in allow:
            _chk(rc, "concatenate")
        return rc


# -- free functions on explicit buffers -------------------------------------------------

def metricise(cfg, raw):
    raw = np.ascontiguousarray(raw, np.uint16)
    out = np.zeros(raw.shape, np.float32)
    lib().smo_metricise(C.byref(cfg), _ptr(raw), _ptr(out))
    return out


def filter_depth(cfg, d, sem, thr):
    d = np.ascontiguousarray(d, np.float32)
    sem = np.ascontiguousarray(sem, np.uint8)
    out = np.zeros(d.shape, np.float32)
    lib().smo_filter_depth(C.byref(cfg), _ptr(d), _ptr(sem), thr, _ptr(out))
    return out


def smooth_depth(cfg, d, sem):
    d = np.ascontiguousarray(d, np.float32)
    sem = np.ascontiguousarray(sem, np.uint8)
    out = np.zeros(d.shape, np.float32)
    lib().smo_smooth_depth(C.byref(cfg), _ptr(d), _ptr(sem), _ptr(out))
    return out


def remove_movings(cfg, d, sem, last, t_c2l):
    d = np.ascontiguousarray(d, np.float32)
    sem = np.ascontiguousarray(sem, np.uint8)
    last = np.ascontiguousarray(last, np.float32)
    t = np.ascontiguousarray(t_c2l, np.float32)
    out = np.zeros(d.shape, np.float32)
    lib().smo_remove_movings(C.byref(cfg), _ptr(d), _ptr(sem), _ptr(last), _ptr(t), _ptr(out))
    return out


def invert4(m):
    m = np.ascontiguousarray(m, np.float32).reshape(16)
    out = np.zeros(16, np.float32)
    fp = C.POINTER(C.c_float)
    lib().smo_invert4(m.ctypes.data_as(fp), out.ctypes.data_as(fp))
    return out
