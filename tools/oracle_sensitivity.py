#!/usr/bin/env python3
"""How far may a real GL run differ from the oracle?  (north_star: "within a stated float
tolerance, surfel count exact".)

The reference pins no numeric result of the hot path and its GL implementation cannot run here, so the
oracle's bits rest on the free choices listed in DESIGN.md 2.  This tool measures what each choice is worth: it
builds oracle VARIANTS that replace one choice by another behaviour a GL driver may legitimately show

    fma         a*b+c contracted to one fused multiply-add      (-ffp-contract=fast -mfma)
    rcp         a/b evaluated as a*(1/b)                         (-DSMO_VAR_RCP)
    libm        libm acosf/expf instead of the fixed kernels     (-DSMO_VAR_LIBM)
    d24_trunc   24-bit depth by truncation, not round-half-up    (-DSMO_VAR_D24_TRUNC)
    inv_double  pose.inverse() in double, rounded once           (-DSMO_VAR_INV_DOUBLE)
    all         everything above together

runs BASELINE configs[0], configs[1] and a 2 M-surfel 1920x1080 case through the contract build and every variant, and
reports per-frame counter deltas and per-field differences of the final models.  Test infrastructure only: it uses
oracle/ and synthetic frames, never the product.

    python tools/oracle_sensitivity.py [--out profiles/oracle_sensitivity] [--quick]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle_lib as ol  # noqa: E402  (config struct + counts struct only; the libraries are loaded here)
from surfelmapping_amd import synth  # noqa: E402

BASE_FLAGS = ["-O2", "-std=c11", "-fPIC", "-fno-fast-math", "-fno-unsafe-math-optimizations", "-D_DEFAULT_SOURCE"]
VARIANTS = {
    "contract": ["-ffp-contract=off"],
    "fma": ["-ffp-contract=fast", "-mfma"],
    "rcp": ["-ffp-contract=off", "-DSMO_VAR_RCP"],
    "libm": ["-ffp-contract=off", "-DSMO_VAR_LIBM"],
    "d24_trunc": ["-ffp-contract=off", "-DSMO_VAR_D24_TRUNC"],
    "inv_double": ["-ffp-contract=off", "-DSMO_VAR_INV_DOUBLE"],
    "all": ["-ffp-contract=fast", "-mfma", "-DSMO_VAR_RCP", "-DSMO_VAR_LIBM", "-DSMO_VAR_D24_TRUNC", "-DSMO_VAR_INV_DOUBLE"],
}
COUNT_KEYS = ("count", "offset", "conflict_count", "fused_count", "unstable_count", "visible_count")


def build_variant(name: str, outdir: str) -> str:
    so = os.path.join(outdir, f"libsmo_{name}.so")
    src = os.path.join(ROOT, "oracle", "smo.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", *BASE_FLAGS, *VARIANTS[name], "-shared", "-o", so, src, "-lm"])
    return so


class Lib:
    def __init__(self, path):
        L = C.CDLL(path)
        L.smo_create.restype = C.c_void_p
        L.smo_create.argtypes = [C.POINTER(ol.SmoConfig)]
        L.smo_destroy.argtypes = [C.c_void_p]
        L.smo_process_frame.argtypes = [C.c_void_p] * 5
        L.smo_get_counts.argtypes = [C.c_void_p, C.POINTER(ol.SmoCounts)]
        L.smo_download_model.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.smo_upload_model.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.smo_set_tick.argtypes = [C.c_void_p, C.c_int32]
        self.L = L


class Run:
    def __init__(self, lib: Lib, cfg):
        self.L = lib.L
        self.h = self.L.smo_create(C.byref(cfg))
        assert self.h

    def frame(self, rgb, depth, sem, pose):
        a = [np.ascontiguousarray(rgb, np.uint8), np.ascontiguousarray(depth, np.uint16), np.ascontiguousarray(sem, np.uint8),
             np.ascontiguousarray(pose, np.float32)]
        rc = self.L.smo_process_frame(self.h, *[x.ctypes.data_as(C.c_void_p) for x in a])
        assert rc == 0, rc

    def counts(self):
        c = ol.SmoCounts()
        self.L.smo_get_counts(self.h, C.byref(c))
        return c.as_dict()

    def model(self):
        n = C.c_uint32()
        self.L.smo_download_model(self.h, None, 0, C.byref(n))
        out = np.zeros((n.value, 12), np.float32)
        self.L.smo_download_model(self.h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
        return out

    def upload(self, m, tick):
        m = np.ascontiguousarray(m, np.float32)
        assert self.L.smo_upload_model(self.h, m.ctypes.data_as(C.c_void_p), m.shape[0]) == 0
        self.L.smo_set_tick(self.h, tick)

    def close(self):
        self.L.smo_destroy(self.h)


def ulps(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """distance in units in the last place between two float32 arrays (same sign assumed where it matters)"""
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def compare_models(ref: np.ndarray, var: np.ndarray) -> dict:
    """Match surfels of the two final models (same creation frame, same colour+class word, nearest position within
    5 mm) and report field differences over the matched pairs."""
    from scipy.spatial import cKDTree
    out = {"n_ref": int(ref.shape[0]), "n_var": int(var.shape[0])}
    if ref.shape[0] == 0 or var.shape[0] == 0:
        return out
    if ref.shape == var.shape and np.array_equal(ref.view(np.uint32), var.view(np.uint32)):
        out.update(matched=int(ref.shape[0]), bit_identical=int(ref.shape[0]), unmatched_ref=0, unmatched_var=0)
        for k in ("pos_max_m", "pos_max_ulp", "normal_max_abs", "radius_max_ulp", "conf_diff_surfels", "time_diff_surfels"):
            out[k] = 0
        return out
    # pairs: nearest neighbour within 0.5 mm (a fifth of the closest pixel spacing in these scenes), mutual, created by
    # the same frame.  A surfel without such a partner exists in only one of the two runs (culled / fused differently).
    fin = np.isfinite(var[:, :3]).all(axis=1)
    finr = np.isfinite(ref[:, :3]).all(axis=1)
    vidx, ridx = np.nonzero(fin)[0], np.nonzero(finr)[0]
    tree_v, tree_r = cKDTree(var[fin, :3].astype(np.float64)), cKDTree(ref[finr, :3].astype(np.float64))
    d, j = tree_v.query(ref[finr, :3].astype(np.float64), k=1, distance_upper_bound=0.0005)
    ok = np.isfinite(d)
    ri, vi = ridx[ok], vidx[j[ok]]
    _, jb = tree_r.query(var[vi, :3].astype(np.float64), k=1)
    mutual = ridx[jb] == ri
    ri, vi = ri[mutual], vi[mutual]
    same_init = ref[ri, 6] == var[vi, 6]
    ri, vi = ri[same_init], vi[same_init]
    a, b = ref[ri], var[vi]
    out["matched"] = int(ri.shape[0])
    out["unmatched_ref"] = int(ref.shape[0] - ri.shape[0])
    out["unmatched_var"] = int(var.shape[0] - ri.shape[0])
    out["bit_identical"] = int((a.view(np.uint32) == b.view(np.uint32)).all(axis=1).sum())
    if ri.shape[0] == 0:
        return out
    out["pos_max_m"] = float(np.abs(a[:, :3].astype(np.float64) - b[:, :3]).max())
    out["pos_max_ulp"] = int(ulps(a[:, :3], b[:, :3]).max())
    out["pos_p99_ulp"] = int(np.percentile(ulps(a[:, :3], b[:, :3]).max(axis=1), 99))
    out["normal_max_abs"] = float(np.abs(a[:, 8:11].astype(np.float64) - b[:, 8:11]).max())
    out["radius_max_ulp"] = int(ulps(a[:, 11], b[:, 11]).max())
    out["conf_diff_surfels"] = int((a[:, 3] != b[:, 3]).sum())
    out["time_diff_surfels"] = int((a[:, 7] != b[:, 7]).sum())
    out["colour_diff_surfels"] = int((a[:, 4].view(np.uint32) != b[:, 4].view(np.uint32)).sum())
    return out


def workloads(quick: bool):
    """name -> (config kwargs, frames, optional seeded model)"""
    w = {}
    # BASELINE configs[0]: single 640x480 frame, identity pose: call 1 reference frame, call 2 all new, call 3 all fuse
    seq = synth.make_sequence(synth.VGA, [synth.pose_matrix(0, 0, 0)] * 3, seed=2)
    w["configs0_vga_identity_3calls"] = (dict(**synth.VGA, preprocess=0), seq, None)
    # BASELINE configs[1]: KITTI-shaped street sequence, the bench's frames (noise 15 mm) and the noise-free ones
    n = 12 if quick else 24
    w["configs1_kitti_noise15mm"] = (dict(**synth.KITTI, preprocess=0), synth.make_sequence(synth.KITTI, synth.kitti_trajectory(n), seed=1, noise_mm=15.0), None)
    w["configs1_kitti_noisefree"] = (dict(**synth.KITTI, preprocess=0), synth.make_sequence(synth.KITTI, synth.kitti_trajectory(n), seed=1), None)
    w["configs1_kitti_noise15mm_preprocess"] = (dict(**synth.KITTI, preprocess=1), synth.make_sequence(synth.KITTI, synth.kitti_trajectory(n), seed=1, noise_mm=15.0), None)
    # a static camera over a slanted scene (yawed pose): every re-observation sits exactly on the fuse threshold 0.0
    stat = synth.make_sequence(synth.KITTI, [synth.pose_matrix(0.3, 0.0, 2.0, 7.0)] * (4 if quick else 6), seed=1)
    w["static_camera_yawed_kitti"] = (dict(**synth.KITTI, preprocess=0), stat, None)
    # 1920x1080, model pre-seeded with 2 M surfels (the HBM-stress shape at a size one CPU core finishes)
    m = synth.seeded_model(500_000 if quick else 2_000_000, tick=300, seed=1)
    w["hd_2M_seeded"] = (dict(**synth.HD, preprocess=0, conflict_cap=0, max_sqrt_vertices=3000),
                         synth.make_sequence(synth.HD, synth.kitti_trajectory(3 if quick else 5), seed=1, noise_mm=15.0), (m, 300))
    return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "oracle_sensitivity"))
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    vdir = os.path.join(ROOT, "oracle", "_variants")
    os.makedirs(vdir, exist_ok=True)
    libs = {name: Lib(build_variant(name, vdir)) for name in VARIANTS}
    report = {"variants": {k: " ".join(v) for k, v in VARIANTS.items()}, "workloads": {}}
    for wname, (ckw, frames, seeded) in workloads(args.quick).items():
        t0 = time.time()
        cfg = ol.make_config(ckw.pop("width"), ckw.pop("height"), ckw.pop("fx"), ckw.pop("fy"), ckw.pop("cx"), ckw.pop("cy"), **ckw)
        runs = {}
        for vname, lib in libs.items():
            r = Run(lib, cfg)
            if seeded is not None:
                r.upload(*seeded)
            per_frame = []
            for fr in frames:
                r.frame(*fr)
                per_frame.append(r.counts())
            runs[vname] = (per_frame, r.model())
            r.close()
        ref_pf, ref_model = runs["contract"]
        wrep = {"frames": len(frames), "contract_final": {k: ref_pf[-1][k] for k in COUNT_KEYS}, "variants": {}}
        for vname, (pf, model) in runs.items():
            if vname == "contract":
                continue
            dmax = {k: max(abs(a[k] - b[k]) for a, b in zip(pf, ref_pf)) for k in COUNT_KEYS}
            dfin = {k: pf[-1][k] - ref_pf[-1][k] for k in COUNT_KEYS}
            first = next((i for i, (a, b) in enumerate(zip(pf, ref_pf)) if any(a[k] != b[k] for k in COUNT_KEYS)), None)
            wrep["variants"][vname] = {"count_delta_final": dfin, "count_delta_max_over_frames": dmax,
                                       "first_frame_with_a_count_difference": first, "model": compare_models(ref_model, model)}
        report["workloads"][wname] = wrep
        print(f"{wname}: {time.time() - t0:.1f} s", file=sys.stderr)
    with open(args.out + ".json", "w") as f:
        json.dump(report, f, indent=1)
    write_markdown(report, args.out + ".md")
    print(args.out + ".md")


def write_markdown(rep, path):
    L = []
    A = L.append
    A("# Oracle sensitivity: what the oracle's free choices are worth\n")
    A("Generated by `tools/oracle_sensitivity.py` (CPU only; oracle variants built from `oracle/smo.c` with the switches below).")
    A("Every number compares a VARIANT of the oracle with the CONTRACT build (`-ffp-contract=off`, no switch) on identical")
    A("synthetic inputs.  The HIP path is bit-identical to the contract build (tests/), so these are also the distances")
    A("between the product and a hypothetical GL run that differs from the contract in that one behaviour.\n")
    A("| variant | build |\n|---|---|")
    for k, v in rep["variants"].items():
        A(f"| `{k}` | `{v}` |")
    A("")
    for wname, w in rep["workloads"].items():
        c = w["contract_final"]
        A(f"## {wname}  ({w['frames']} calls)\n")
        A(f"contract build, after the last call: count {c['count']}, conflicts {c['conflict_count']}, fused {c['fused_count']}, "
          f"new {c['unstable_count']}, drawn into the index map {c['visible_count']}\n")
        A("| variant | Δcount final (max over frames) | Δconflicts max | Δfused max | Δnew max | first differing call | matched surfels | bit-identical | max Δpos [m] (ulp, p99 ulp) | max Δnormal | radius max ulp | conf differs |")
        A("|---|---|---|---|---|---|---|---|---|---|---|---|")
        for vname, v in w["variants"].items():
            m, dm, df = v["model"], v["count_delta_max_over_frames"], v["count_delta_final"]
            A(f"| `{vname}` | {df['count']:+d} ({dm['count']}) | {dm['conflict_count']} | {dm['fused_count']} | {dm['unstable_count']} | "
              f"{v['first_frame_with_a_count_difference']} | {m.get('matched', 0)} of {m['n_ref']} | {m.get('bit_identical', 0)} | "
              f"{m.get('pos_max_m', 0):.2e} ({m.get('pos_max_ulp', 0)}, {m.get('pos_p99_ulp', 0)}) | {m.get('normal_max_abs', 0):.1e} | "
              f"{m.get('radius_max_ulp', 0)} | {m.get('conf_diff_surfels', 0)} |")
        A("")
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")


if __name__ == "__main__":
    main()
