"""ctypes binding of the C-ABI in include/sm_c_api.h (libsurfelmapping_hip.so).

This is plumbing only: every call goes straight to the HIP library.  There is NO CPU
fallback -- if the shared library is missing or no GPU is visible, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SM_HIP_LIB") or os.path.join(_PKG, "libsurfelmapping_hip.so")      # (SM_HIP_LIB: another build of the core, for A/B runs)
API_VERSION = 4                # SM_API_VERSION of include/sm_c_api.h

SM_OK, SM_E_ARG, SM_E_CAPACITY, SM_E_UNSUPPORTED, SM_E_HIP, SM_E_NO_DEVICE = 0, -1, -2, -3, -4, -5
TEX_DEPTH_METRIC, TEX_DEPTH_FILTERED, TEX_LAST = 0, 1, 2

# every extern "C" symbol include/sm_c_api.h declares
SYMBOLS = (
    "sm_api_version", "sm_last_error", "sm_default_config", "sm_create", "sm_destroy",
    "sm_process_frame", "sm_process_frame_device", "sm_process_frame_async",
    "sm_inputs_consumed", "sm_host_alloc", "sm_host_alloc_frame", "sm_host_free", "sm_debug_slow_frames", "sm_sync", "sm_clean_points", "sm_clean_points_ex", "sm_clean_points_cb", "sm_reset",
    "sm_get_counts", "sm_download_model_aos", "sm_upload_model_aos", "sm_save_map", "sm_load_map",
    "sm_download_index_map", "sm_download_raw_cloud", "sm_download_depth", "sm_render_image", "sm_set_frame", "sm_set_tick",
    "sm_stage_conflict", "sm_stage_cull", "sm_stage_splat", "sm_stage_associate_fuse",
    "sm_stage_timings", "sm_read_frame_log", "sm_device_alloc", "sm_device_free", "sm_device_upload",
    "sm_export_model_device", "sm_append_model_aos_device", "sm_device_download",
    "sm_shard_stream_configure", "sm_shard_set_collective", "sm_shard_rccl_unique_id", "sm_shard_rccl_init",
    "sm_shard_rccl_finalize", "sm_shard_rccl_nranks", "sm_shard_frame_device", "sm_shard_frame", "sm_shard_compact", "sm_shard_export_dense_device",
    "sm_gpu_process_count", "sm_rig_configure", "sm_rig_consolidate", "sm_rig_consolidate_step",
)

SM_COLL_SUM, SM_COLL_MIN, SM_COLL_GATHER = 0, 1, 2
# int fn(void *user, const void *send, void *recv, size_t count_u64, int op, void *hip_stream)
COLLECTIVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
# long long fn(void *user, uint32_t local_conflicts)
CAP_FN = C.CFUNCTYPE(C.c_longlong, C.c_void_p, C.c_uint32)


class SmConfig(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("near_clip", C.c_float), ("far_clip", C.c_float), ("fuse_thresh", C.c_float),
        ("max_sqrt_vertices", C.c_int32), ("time_delta", C.c_int32),
        ("stereo_border", C.c_float), ("preprocess", C.c_int32), ("conflict_cap", C.c_int32),
        ("device", C.c_int32), ("enable_timing", C.c_int32), ("disable_tile_bounds", C.c_int32),
        ("compact_period", C.c_int32),
    ]


class SmCounts(C.Structure):
    _fields_ = [
        ("count", C.c_uint32), ("offset", C.c_uint32), ("data_count", C.c_uint32),
        ("conflict_count", C.c_uint32), ("unstable_count", C.c_uint32),
        ("fused_count", C.c_uint32), ("visible_count", C.c_uint32), ("tick", C.c_int32),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class SmTimings(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "preprocess", "conflict", "index_map", "data_association", "concatenate", "run",
        "k_prep", "k_conflict", "k_scan_cull", "k_compact", "k_associate", "k_scan_new",
        "k_append")] + [("frames", C.c_uint32), ("event_overhead", C.c_float), ("k_compact_own", C.c_float),
                        ("k_cull_lazy", C.c_float), ("frames_compact", C.c_uint32)] + [(n, C.c_float) for n in (
        "k_surfel_pass", "k_pass_fixup", "k_conflict_own", "k_associate_direct", "k_associate_own", "k_append_own")] + [
        ("frames_one_pass", C.c_uint32), ("frames_direct", C.c_uint32), ("k_assoc_prep", C.c_float), ("k_prep_own", C.c_float),
        ("frames_merged", C.c_uint32), ("frames_assoc_alone", C.c_uint32), ("k_scan_own", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


FRAME_LOG_LEN = 1024
FRAME_LOG_DTYPE = np.dtype([(n, np.uint32) for n in (
    "tick", "n_before", "n_after_cull", "n_kill", "conflict_count", "visible_count",
    "fused_count", "unstable_count", "n_static", "n_conf_skipped", "n_splat_skipped", "n_slots")])


class SurfelMapError(RuntimeError):
    def __init__(self, what, rc, detail=""):
        super().__init__(f"{what} failed: rc={rc} {detail}".strip())
        self.rc = rc


_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch's ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7) and ask
    for it by the name "libamdhip64.so"; the HIP core asks for "libamdhip64.so.7".  If the core is loaded first, ROCm's copy
    comes in and torch later loads its own as well -- two runtimes, and RCCL (torch's) cannot use the core's device memory
    (sm_create / sm_export_model_device refuse with both paths).  Loading torch's copy first makes either import order work:
    the core's request matches its soname.  No torch, or SM_NO_TORCH_HIP_PRELOAD=1: nothing is done."""
    if os.environ.get("SM_NO_TORCH_HIP_PRELOAD") == "1":
        return
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return                                   # torch's runtime is loaded already
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if not spec or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _choose_rccl():
    """RCCL for the in-stream sharded mode is bound at run time by the core.  When PyTorch is installed this module has made
    PyTorch's bundled HIP runtime the process's runtime (above); name PyTorch's bundled RCCL too, so that both come from one
    build.  SM_RCCL_LIB set by the user wins; without PyTorch the core takes ROCm's librccl."""
    if os.environ.get("SM_RCCL_LIB") or os.environ.get("SM_NO_TORCH_HIP_PRELOAD") == "1":
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
        if os.path.exists(cand):
            os.environ["SM_RCCL_LIB"] = cand


def rccl_unique_id() -> bytes:
    """128-byte RCCL id made by one rank and handed to all (sm_shard_rccl_unique_id)"""
    _choose_rccl()
    L = load()
    buf = C.create_string_buffer(128)
    rc = L.sm_shard_rccl_unique_id(buf)
    if rc:
        raise SurfelMapError("sm_shard_rccl_unique_id", rc, L.sm_last_error().decode())
    return buf.raw


def load():
    """dlopen the HIP library (no compute).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C surfelmapping_amd/csrc). surfelmapping_amd has no CPU fallback.")
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, u32p = C.c_void_p, C.POINTER(C.c_uint32)
    L.sm_api_version.restype = C.c_int
    if L.sm_api_version() != API_VERSION:     # a stale build: the struct layouts below would not match
        raise ImportError(f"{LIB_PATH} has API version {L.sm_api_version()}, this binding expects {API_VERSION}: rebuild it "
                          "(python -c 'import __graft_entry__ as g; g.build()')")
    L.sm_last_error.restype = C.c_char_p
    L.sm_default_config.argtypes = [C.POINTER(SmConfig), C.c_int, C.c_int] + [C.c_float] * 4
    L.sm_create.restype = vp
    L.sm_create.argtypes = [C.POINTER(SmConfig)]
    L.sm_destroy.restype = None
    L.sm_destroy.argtypes = [vp]
    L.sm_process_frame.argtypes = [vp, vp, vp, vp, vp]
    L.sm_process_frame_device.argtypes = [vp, vp, vp, vp, vp]
    L.sm_process_frame_async.argtypes = [vp, vp, vp, vp, vp]
    L.sm_inputs_consumed.argtypes = [vp]
    L.sm_host_alloc.restype = vp
    L.sm_host_alloc.argtypes = [vp, C.c_size_t]
    L.sm_host_free.argtypes = [vp, vp]
    L.sm_host_alloc_frame.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.sm_debug_slow_frames.argtypes = [vp, u32p]
    L.sm_sync.argtypes = [vp]
    L.sm_clean_points.argtypes = [vp, vp, vp, vp]
    L.sm_clean_points_ex.argtypes = [vp, vp, vp, vp, C.c_int]
    L.sm_clean_points_cb.argtypes = [vp, vp, vp, vp, C.c_int, CAP_FN, vp]
    L.sm_reset.argtypes = [vp]
    L.sm_get_counts.argtypes = [vp, C.POINTER(SmCounts)]
    L.sm_download_model_aos.argtypes = [vp, vp, C.c_uint32, u32p]
    L.sm_upload_model_aos.argtypes = [vp, vp, C.c_uint32]
    L.sm_save_map.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32]
    L.sm_load_map.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.sm_download_index_map.argtypes = [vp, vp, vp, vp, vp]
    L.sm_download_raw_cloud.argtypes = [vp, vp, C.c_uint32, u32p]
    L.sm_download_depth.argtypes = [vp, C.c_int, vp]
    L.sm_render_image.argtypes = [vp, vp, C.c_int, C.c_int] + [C.c_float] * 4 + [vp, vp]
    L.sm_set_frame.argtypes = [vp, vp, vp, vp]
    L.sm_set_tick.argtypes = [vp, C.c_int32]
    L.sm_stage_conflict.argtypes = [vp, vp, C.c_float, C.c_float, C.c_float, C.c_int]
    L.sm_stage_cull.argtypes = [vp]
    L.sm_stage_splat.argtypes = [vp, vp, C.c_int32, C.c_float, C.c_int32]
    L.sm_stage_associate_fuse.argtypes = [vp, vp, C.c_int32, C.c_float, C.c_float]
    L.sm_stage_timings.argtypes = [vp, C.POINTER(SmTimings)]
    L.sm_read_frame_log.argtypes = [vp, vp, C.c_uint32, u32p]
    L.sm_device_alloc.restype = vp
    L.sm_device_alloc.argtypes = [vp, C.c_size_t]
    L.sm_device_free.argtypes = [vp, vp]
    L.sm_device_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.sm_export_model_device.argtypes = [vp, C.POINTER(vp), u32p]
    L.sm_append_model_aos_device.argtypes = [vp, vp, C.c_uint32]
    L.sm_device_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.sm_shard_stream_configure.argtypes = [vp, C.c_int, C.c_int]
    L.sm_shard_set_collective.argtypes = [vp, COLLECTIVE_FN, vp]
    L.sm_shard_rccl_unique_id.argtypes = [vp]
    L.sm_shard_rccl_init.argtypes = [vp, vp]
    L.sm_shard_rccl_finalize.argtypes = [vp]
    L.sm_shard_rccl_nranks.argtypes = [vp]
    L.sm_shard_frame_device.argtypes = [vp, vp, vp, vp, vp]
    L.sm_shard_frame.argtypes = [vp, vp, vp, vp, vp]
    L.sm_shard_compact.argtypes = [vp]
    L.sm_shard_export_dense_device.argtypes = [vp, C.POINTER(vp), u32p]
    L.sm_gpu_process_count.argtypes = [vp]
    L.sm_rig_configure.argtypes = [vp, C.c_int, C.c_int]
    L.sm_rig_consolidate.argtypes = [vp, vp, vp, vp, vp, u32p, u32p]
    L.sm_rig_consolidate_step.argtypes = [vp, vp, vp, vp, vp, u32p, u32p]
    for name in SYMBOLS:
        getattr(L, name)          # AttributeError here = the library does not match the header
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_config(width, height, fx, fy, cx, cy, **over) -> SmConfig:
    c = SmConfig()
    load().sm_default_config(C.byref(c), width, height, fx, fy, cx, cy)
    for k, v in over.items():
        if not hasattr(c, k):
            raise KeyError(k)
        setattr(c, k, v)
    return c


class SurfelMap:
    """Host-side handle mirroring SurfelMapping / GlobalModel / IndexMap of the reference
    (src/SurfelMapping.h:31-96, src/GlobalModel.h:22-120, src/IndexMap.h:34-88)."""

    def __init__(self, cfg: SmConfig):
        self._L = load()
        self.cfg = cfg
        self.W, self.H = cfg.width, cfg.height
        self.P = self.W * self.H
        self._h = self._L.sm_create(C.byref(cfg))
        if not self._h:
            raise SurfelMapError("sm_create", SM_E_NO_DEVICE, self._L.sm_last_error().decode())

    def _chk(self, rc, what, allow=(0,)):
        if rc not in allow:
            raise SurfelMapError(what, rc, self._L.sm_last_error().decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._L.sm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- SurfelMapping
    def process_frame(self, rgb, depth, sem, pose, allow=(0,)):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = None if depth is None else np.ascontiguousarray(depth, np.uint16)
        sem = None if sem is None else np.ascontiguousarray(sem, np.uint8)
        pose = np.ascontiguousarray(pose, np.float32)
        return self._chk(self._L.sm_process_frame(self._h, _ptr(rgb), _ptr(depth), _ptr(sem), _ptr(pose)),
                         "sm_process_frame", allow)

    def process_frame_device(self, d_rgb, d_depth, d_sem, pose):
        pose = np.ascontiguousarray(pose, np.float32)
        return self._chk(self._L.sm_process_frame_device(self._h, d_rgb, d_depth, d_sem, _ptr(pose)),
                         "sm_process_frame_device")

    def process_frame_async(self, rgb, depth, sem, pose):
        """host arrays, no host wait: the copy of this frame overlaps the previous frame (sm_process_frame_async).  Arrays made by
        host_array() (pinned, owned by the context) are read in place until inputs_consumed() / sync(); others are staged inside
        the call."""
        assert rgb.dtype == np.uint8 and rgb.flags.c_contiguous
        pose = np.ascontiguousarray(pose, np.float32)
        return self._chk(self._L.sm_process_frame_async(self._h, _ptr(rgb), _ptr(depth), _ptr(sem), _ptr(pose)), "sm_process_frame_async")

    def host_array(self, shape, dtype) -> np.ndarray:
        """a numpy array in pinned host memory owned by the context (sm_host_alloc): the fastest source for process_frame_async"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self._L.sm_host_alloc(self._h, n)
        if not p:
            raise SurfelMapError("sm_host_alloc", SM_E_HIP, self._L.sm_last_error().decode())
        buf = (C.c_ubyte * n).from_address(p)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def host_frame(self):
        """(rgb (H, W, 3) u8, depth (H, W) u16, semantic (H, W) u8): numpy views of ONE pinned block (sm_host_alloc_frame) --
        process_frame_async copies such a frame with a single transfer"""
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self._L.sm_host_alloc_frame(self._h, C.byref(a), C.byref(b), C.byref(c)), "sm_host_alloc_frame")
        H, W = self.H, self.W
        rgb = np.frombuffer((C.c_ubyte * (H * W * 3)).from_address(a.value), dtype=np.uint8).reshape(H, W, 3)
        dep = np.frombuffer((C.c_ubyte * (H * W * 2)).from_address(b.value), dtype=np.uint16).reshape(H, W)
        sem = np.frombuffer((C.c_ubyte * (H * W)).from_address(c.value), dtype=np.uint8).reshape(H, W)
        return rgb, dep, sem

    def debug_slow_frames(self) -> int:
        """frames that took the rare path of the two-launch frame so far (diagnostic; synchronises)"""
        n = C.c_uint32()
        self._chk(self._L.sm_debug_slow_frames(self._h, C.byref(n)), "sm_debug_slow_frames")
        return int(n.value)

    def inputs_consumed(self):
        self._chk(self._L.sm_inputs_consumed(self._h), "sm_inputs_consumed")

    def sync(self, allow=(0,)):
        return self._chk(self._L.sm_sync(self._h), "sm_sync", allow)

    def clean_points(self, depth, sem, pose):
        depth = np.ascontiguousarray(depth, np.uint16)
        sem = np.ascontiguousarray(sem, np.uint8)
        pose = np.ascontiguousarray(pose, np.float32)
        self._chk(self._L.sm_clean_points(self._h, _ptr(depth), _ptr(sem), _ptr(pose)), "sm_clean_points")

    def clean_points_slice(self, depth, sem, pose, exempt_first: bool, cap_hook=None):
        """cleanPoints on a rig slice (surfelmapping_amd/dist.py): the id-0 exemption only where the slice holds the global surfel
        0; cap_hook(local_conflicts) -> how many of them may take effect (the slice's share of the union's W*H conflict
        records), called between the conflict test and the cull"""
        depth = np.ascontiguousarray(depth, np.uint16)
        sem = np.ascontiguousarray(sem, np.uint8)
        pose = np.ascontiguousarray(pose, np.float32)
        if cap_hook is None:
            self._chk(self._L.sm_clean_points_ex(self._h, _ptr(depth), _ptr(sem), _ptr(pose), 1 if exempt_first else 0), "sm_clean_points_ex")
            return
        err = []

        def tramp(user, local):
            try:
                return int(cap_hook(int(local)))
            except Exception as e:                       # never let an exception cross the C frame
                err.append(e)
                return SM_E_HIP
        cb = CAP_FN(tramp)
        rc = self._L.sm_clean_points_cb(self._h, _ptr(depth), _ptr(sem), _ptr(pose), 1 if exempt_first else 0, cb, None)
        if err:
            raise err[0]
        self._chk(rc, "sm_clean_points_cb")

    def reset(self):
        self._chk(self._L.sm_reset(self._h), "sm_reset")

    # -- GlobalModel
    def counts(self) -> dict:
        c = SmCounts()
        self._chk(self._L.sm_get_counts(self._h, C.byref(c)), "sm_get_counts")
        return c.as_dict()

    def download_model(self) -> np.ndarray:
        n = C.c_uint32()
        self._chk(self._L.sm_download_model_aos(self._h, None, 0, C.byref(n)), "sm_download_model_aos")
        out = np.zeros((n.value, 12), np.float32)
        self._chk(self._L.sm_download_model_aos(self._h, _ptr(out), n.value, C.byref(n)), "sm_download_model_aos")
        return out

    def upload_model(self, m):
        m = np.ascontiguousarray(m, np.float32)
        self._chk(self._L.sm_upload_model_aos(self._h, _ptr(m), m.shape[0]), "sm_upload_model_aos")

    def save_map(self, path, start_id=0, end_id=0):
        self._chk(self._L.sm_save_map(self._h, os.fsencode(path), start_id, end_id), "sm_save_map")

    def load_map(self, path):
        a, b = C.c_int32(), C.c_int32()
        self._chk(self._L.sm_load_map(self._h, os.fsencode(path), C.byref(a), C.byref(b)), "sm_load_map")
        return a.value, b.value

    # -- IndexMap
    def download_index_map(self):
        P = self.P
        idx = np.zeros(P, np.int32)
        vc = np.zeros((P, 4), np.float32)
        ct = np.zeros((P, 4), np.float32)
        nr = np.zeros((P, 4), np.float32)
        self._chk(self._L.sm_download_index_map(self._h, _ptr(idx), _ptr(vc), _ptr(ct), _ptr(nr)),
                  "sm_download_index_map")
        return idx, vc, ct, nr

    def download_raw_cloud(self) -> np.ndarray:
        """FeedbackBuffer "RAW": the raw camera-frame surfel cloud of the last processed frame, float32 [n][12]."""
        n = C.c_uint32()
        self._chk(self._L.sm_download_raw_cloud(self._h, None, 0, C.byref(n)), "sm_download_raw_cloud")
        out = np.zeros((n.value, 12), np.float32)
        if n.value:
            self._chk(self._L.sm_download_raw_cloud(self._h, _ptr(out), n.value, C.byref(n)), "sm_download_raw_cloud")
        return out

    def download_depth(self, which=TEX_DEPTH_METRIC):
        out = np.zeros((self.H, self.W), np.float32)
        self._chk(self._L.sm_download_depth(self._h, which, _ptr(out)), "sm_download_depth")
        return out

    def render_image(self, view, w, h, fx, fy, cx, cy):
        """Novel view (GlobalModel::renderImage): (bgr uint8[h][w][3], semantic uint8[h][w] = class + 1)."""
        view = np.ascontiguousarray(view, np.float32)
        bgr = np.zeros((h, w, 3), np.uint8)
        sem = np.zeros((h, w), np.uint8)
        self._chk(self._L.sm_render_image(self._h, _ptr(view), w, h, fx, fy, cx, cy, _ptr(bgr), _ptr(sem)), "sm_render_image")
        return bgr, sem

    # -- per-pass entry points
    def set_frame(self, rgb=None, depth_metric=None, sem=None):
        rgb = None if rgb is None else np.ascontiguousarray(rgb, np.uint8)
        dm = None if depth_metric is None else np.ascontiguousarray(depth_metric, np.float32)
        sem = None if sem is None else np.ascontiguousarray(sem, np.uint8)
        self._chk(self._L.sm_set_frame(self._h, _ptr(rgb), _ptr(dm), _ptr(sem)), "sm_set_frame")

    def set_tick(self, tick):
        self._chk(self._L.sm_set_tick(self._h, tick), "sm_set_tick")

    def stage_conflict(self, pose, min_depth, max_depth, fuse_thresh=0.0, is_clean=0):
        pose = np.ascontiguousarray(pose, np.float32)
        self._chk(self._L.sm_stage_conflict(self._h, _ptr(pose), min_depth, max_depth, fuse_thresh, is_clean),
                  "sm_stage_conflict")

    def stage_cull(self):
        self._chk(self._L.sm_stage_cull(self._h), "sm_stage_cull")

    def stage_splat(self, pose, time, depth_cutoff, time_delta):
        pose = np.ascontiguousarray(pose, np.float32)
        self._chk(self._L.sm_stage_splat(self._h, _ptr(pose), time, depth_cutoff, time_delta), "sm_stage_splat")

    def stage_associate_fuse(self, pose, time, dmin, dmax, allow=(0,)):
        pose = np.ascontiguousarray(pose, np.float32)
        return self._chk(self._L.sm_stage_associate_fuse(self._h, _ptr(pose), time, dmin, dmax),
                         "sm_stage_associate_fuse", allow)

    def timings(self) -> dict:
        t = SmTimings()
        self._chk(self._L.sm_stage_timings(self._h, C.byref(t)), "sm_stage_timings")
        return t.as_dict()

    def read_frame_log(self, n: int = FRAME_LOG_LEN) -> np.ndarray:
        """Newest `n` per-frame counter records written by the device (oldest first)."""
        n = min(n, FRAME_LOG_LEN)
        out = np.zeros(n, FRAME_LOG_DTYPE)
        w = C.c_uint32()
        self._chk(self._L.sm_read_frame_log(self._h, _ptr(out), n, C.byref(w)), "sm_read_frame_log")
        return out[:w.value]

    # -- device staging helpers
    def device_alloc(self, nbytes: int) -> int:
        p = self._L.sm_device_alloc(self._h, nbytes)
        if not p:
            raise SurfelMapError("sm_device_alloc", SM_E_HIP, self._L.sm_last_error().decode())
        return p

    def device_free(self, p: int):
        self._chk(self._L.sm_device_free(self._h, p), "sm_device_free")

    def device_upload(self, dst: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._chk(self._L.sm_device_upload(self._h, dst, _ptr(arr), arr.nbytes), "sm_device_upload")

    def export_model_device(self):
        """(device pointer of an AoS float32[n][12] staging copy of the model, n)."""
        p, n = C.c_void_p(), C.c_uint32()
        self._chk(self._L.sm_export_model_device(self._h, C.byref(p), C.byref(n)), "sm_export_model_device")
        return p.value, n.value

    def append_model_device(self, d_ptr: int, n: int):
        self._chk(self._L.sm_append_model_aos_device(self._h, d_ptr, n), "sm_append_model_aos_device")

    def device_download(self, src: int, nbytes: int, dtype=np.uint8) -> np.ndarray:
        out = np.zeros(nbytes // np.dtype(dtype).itemsize, dtype)
        self._chk(self._L.sm_device_download(self._h, _ptr(out), src, nbytes), "sm_device_download")
        return out

    # -- ONE stream sharded over several GPUs (slot-addressed; the collectives run on the context's stream; surfelmapping_amd/sharded.py)
    def shard_stream_configure(self, rank, world):
        self._chk(self._L.sm_shard_stream_configure(self._h, rank, world), "sm_shard_stream_configure")

    def shard_set_collective(self, fn):
        """fn(send_ptr, recv_ptr, count_u64, op) -> 0: an all-reduce over the ranks with the meaning of sm_collective_fn"""
        def tramp(user, send, recv, count, op, stream):
            try:
                return int(fn(send, recv, count, op) or 0)
            except Exception as e:                       # never let an exception cross the C frame
                self._coll_error = e
                return SM_E_HIP
        self._coll_cb = COLLECTIVE_FN(tramp)            # keep the trampoline alive as long as the context
        self._chk(self._L.sm_shard_set_collective(self._h, self._coll_cb, None), "sm_shard_set_collective")

    def shard_rccl_init(self, unique_id: bytes):
        _choose_rccl()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self._L.sm_shard_rccl_init(self._h, buf), "sm_shard_rccl_init")

    def shard_rccl_nranks(self) -> int:
        """ranks of this context's RCCL communicator, as RCCL reports them"""
        n = self._L.sm_shard_rccl_nranks(self._h)
        if n < 0:
            raise SurfelMapError("sm_shard_rccl_nranks", n, self._L.sm_last_error().decode())
        return n

    def shard_rccl_finalize(self):
        self._chk(self._L.sm_shard_rccl_finalize(self._h), "sm_shard_rccl_finalize")

    def shard_frame(self, rgb, depth, sem, pose, allow=(0,)):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = None if depth is None else np.ascontiguousarray(depth, np.uint16)
        sem = None if sem is None else np.ascontiguousarray(sem, np.uint8)
        pose = np.ascontiguousarray(pose, np.float32)
        rc = self._L.sm_shard_frame(self._h, _ptr(rgb), _ptr(depth), _ptr(sem), _ptr(pose))
        if rc and getattr(self, "_coll_error", None) is not None:
            e, self._coll_error = self._coll_error, None
            raise e
        return self._chk(rc, "sm_shard_frame", allow)

    def shard_frame_device(self, d_rgb, d_depth, d_sem, pose):
        pose = np.ascontiguousarray(pose, np.float32)
        return self._chk(self._L.sm_shard_frame_device(self._h, d_rgb, d_depth, d_sem, _ptr(pose)), "sm_shard_frame_device")

    def shard_compact(self):
        self._chk(self._L.sm_shard_compact(self._h), "sm_shard_compact")

    def shard_export_dense_device(self):
        p, n = C.c_void_p(), C.c_uint32()
        self._chk(self._L.sm_shard_export_dense_device(self._h, C.byref(p), C.byref(n)), "sm_shard_export_dense_device")
        return p.value, n.value

    def shard_export_dense(self) -> np.ndarray:
        """this rank's surfels of the compacted union, zeros in the other ranks' slots: (count, 12) float32 (collective)"""
        p, n = self.shard_export_dense_device()
        if n == 0:
            return np.zeros((0, 12), np.float32)
        return self.device_download(p, n * 48, np.float32).reshape(n, 12)

    # -- rig mode (configs[4]): consolidation into a single GlobalModel inside the core
    def rig_configure(self, rank, world):
        self._chk(self._L.sm_rig_configure(self._h, rank, world), "sm_rig_configure")
        self._rig_world = world

    def rig_consolidate(self, depth, sem, pose, sm_global):
        """collective: -> (surfels in the single GlobalModel now appended to `sm_global`, conflicts per view)"""
        depth = np.ascontiguousarray(depth, np.uint16)
        sem = np.ascontiguousarray(sem, np.uint8)
        pose = np.ascontiguousarray(pose, np.float32)
        per_view = (C.c_uint32 * max(getattr(self, "_rig_world", 1), 1))()
        total = C.c_uint32()
        rc = self._L.sm_rig_consolidate(self._h, _ptr(depth), _ptr(sem), _ptr(pose), sm_global._h, per_view, C.byref(total))
        if rc and getattr(self, "_coll_error", None) is not None:
            e, self._coll_error = self._coll_error, None
            raise e
        self._chk(rc, "sm_rig_consolidate")
        return total.value, [int(x) for x in per_view]

    def rig_consolidate_step(self, depth, sem, pose, sm_global):
        """collective, every K frames: -> (new surfels exchanged in this step, surfels in the single GlobalModel `sm_global` after it)"""
        depth = np.ascontiguousarray(depth, np.uint16)
        sem = np.ascontiguousarray(sem, np.uint8)
        pose = np.ascontiguousarray(pose, np.float32)
        new, tot = C.c_uint32(), C.c_uint32()
        rc = self._L.sm_rig_consolidate_step(self._h, _ptr(depth), _ptr(sem), _ptr(pose), sm_global._h, C.byref(new), C.byref(tot))
        if rc and getattr(self, "_coll_error", None) is not None:
            e, self._coll_error = self._coll_error, None
            raise e
        self._chk(rc, "sm_rig_consolidate_step")
        return new.value, tot.value

    def gpu_process_count(self) -> int:
        """processes with compute queues on this context's GPU per the KFD tables (this one included); -1 if unreadable"""
        return self._L.sm_gpu_process_count(self._h)
