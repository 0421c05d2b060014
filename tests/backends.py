"""Backends the parity tests run against: the CPU oracle (checker) and the HIP product.

`HipAsPasses` lets a test written against the reference's pass names (processConflict,
updateConflict, backMapping, buildModelMap, predictIndices, dataAssociate, updateFuse,
concatenate) drive the product's fused per-pass entry points of include/sm_c_api.h.
"""
import numpy as np
import pytest

import oracle_lib as ol


def gpu_available() -> bool:
    try:
        from surfelmapping_amd import capi
        L = capi.load()
        cfg = capi.make_config(64, 64, 50.0, 50.0, 31.5, 31.5, max_sqrt_vertices=64)
        import ctypes as C
        h = L.sm_create(C.byref(cfg))
        if not h:
            return False
        L.sm_destroy(h)
        return True
    except Exception:
        return False


class HipAsPasses:
    def __init__(self, cfg):
        from surfelmapping_amd import capi
        self.m = capi.SurfelMap(cfg)
        self.cfg = cfg
        self._assoc_args = None

    def __getattr__(self, name):
        return getattr(self.m, name)

    # pass-name adapters -------------------------------------------------------------
    def stage_process_conflict(self, pose, min_depth, max_depth, fuse_thresh=0.0, is_clean=0):
        self.m.stage_conflict(pose, min_depth, max_depth, fuse_thresh, is_clean)
        self._pending = True

    def stage_update_conflict(self):
        pass                                   # in-place decrement, applied by the cull

    def stage_back_mapping(self):
        if getattr(self, "_pending", False):
            self.m.stage_cull()
            self._pending = False

    def stage_build_model_map(self):
        pass                                   # no mirror textures in the product

    def stage_predict_indices(self, pose, time, depth_cutoff, time_delta):
        self.m.stage_splat(pose, time, depth_cutoff, time_delta)

    def stage_data_associate(self, pose, time, dmin, dmax):
        self.m.stage_associate_fuse(pose, time, dmin, dmax)

    def stage_update_fuse(self):
        pass

    def stage_concatenate(self, allow=(0,)):
        return 0


def make(backend: str, W, H, fx, fy, cx, cy, **over):
    if backend == "oracle":
        over = {k: v for k, v in over.items() if k not in ("disable_tile_bounds", "device", "enable_timing", "compact_period")}
        return ol.Oracle(ol.make_config(W, H, fx, fy, cx, cy, **over))
    from surfelmapping_amd import capi
    over.setdefault("max_sqrt_vertices", 1000)
    return HipAsPasses(capi.make_config(W, H, fx, fy, cx, cy, **over))


BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


def assert_models_equal(a: np.ndarray, b: np.ndarray, what=""):
    """Bit-exact comparison of two AoS models (NaN-safe: compares the raw 32-bit words)."""
    assert a.shape == b.shape, f"{what}: surfel count {a.shape[0]} vs {b.shape[0]}"
    au, bu = a.view(np.uint32), b.view(np.uint32)
    if not np.array_equal(au, bu):
        bad = np.argwhere(au != bu)
        k, f = bad[0]
        raise AssertionError(f"{what}: {len(bad)} words differ; first at surfel {k} field {f}: "
                             f"{a[k]} vs {b[k]}")
