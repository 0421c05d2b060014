#!/usr/bin/env python3
"""CPU study behind DESIGN.md's tile-culling and lane-efficiency figures (no GPU: the oracle's model, numpy).

Runs the bench workload (BASELINE configs[1]) through the oracle, and for the last frames takes the model as it stands
before the frame -- dense creation order, 1024 consecutive surfels = one tile, which approximates the GPU's slot order --
and reports
  * how many surfels / 64-slot words / tiles hold a surfel the exact view tests accept (what the pass has to touch),
  * how many tiles three box tests admit: round 1's (side planes only for boxes entirely in front of the camera), the
    current one (side planes for every box: sm_kernels.h plane_guard), and the current one plus a second box in axes
    rotated by 45 degrees about the vertical (costed in DESIGN.md 8, not built).
usage: tools/tile_cull_study.py [frames=24]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol                      # noqa: E402
from surfelmapping_amd import synth          # noqa: E402
import bench                                 # noqa: E402

NF = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cam = synth.KITTI
frames = bench.make_frames(cam, NF, 1, 15.0, 8)
cfg = ol.make_config(cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], preprocess=0)
o = ol.Oracle(cfg)
W, H, fx, fy, cx, cy = cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"]


def planes(c):
    x, y, z = c[..., 0], c[..., 1], c[..., 2]
    return ((fx * x + (cx - W - 2) * z > 0).all(1) | (fx * x + (cx + 2) * z < 0).all(1) |
            (fy * y + (cy - H - 2) * z > 0).all(1) | (fy * y + (cy + 2) * z < 0).all(1))


for k, (rgb, depth, sem, pose) in enumerate(frames):
    if k >= NF - 3:
        m = o.download_model()
        N = len(m)
        P = np.asarray(pose, dtype=np.float64).reshape(4, 4)
        T = np.linalg.inv(P.T if abs(P[3, :3]).sum() > 1e-6 else P)          # world -> camera
        pos = m[:, :3].astype(np.float64)
        pc = pos @ T[:3, :3].T + T[:3, 3]
        z = pc[:, 2]
        with np.errstate(all="ignore"):
            u = fx * pc[:, 0] / z + cx
            v = fy * pc[:, 1] / z + cy
        inv = (z > 0) & (z < cfg.far_clip * 1.5) & (u >= 0) & (u <= W) & (v >= 0) & (v <= H)
        pad = (-N) % 1024
        a = np.concatenate([inv, np.zeros(pad, bool)])
        words, tiles = a.reshape(-1, 64), a.reshape(-1, 1024)
        need = tiles.any(1)
        pp = np.concatenate([pos, np.repeat(pos[-1:], pad, 0)]).reshape(-1, 1024, 3)

        def corners(lo, hi):
            return np.stack([np.where(np.array([(c >> b) & 1 for b in range(3)], bool), hi, lo) for c in range(8)], 1)

        cw = corners(pp.min(1), pp.max(1)) @ T[:3, :3].T + T[:3, 3]
        zout = (cw[..., 2].max(1) < -0.01) | (cw[..., 2].min(1) > cfg.far_clip + 0.01)
        old = ~(zout | ((cw[..., 2].min(1) > 1e-3) & planes(cw)))
        new = ~(zout | planes(cw))
        ab = np.stack([pp[..., 0] + pp[..., 2], pp[..., 0] - pp[..., 2], pp[..., 1]], -1)
        c2 = corners(ab.min(1), ab.max(1))
        c2w = np.stack([(c2[..., 0] + c2[..., 1]) / 2, c2[..., 2], (c2[..., 0] - c2[..., 1]) / 2], -1) @ T[:3, :3].T + T[:3, 3]
        both = new & ~(((c2w[..., 2].max(1) < -0.01) | (c2w[..., 2].min(1) > cfg.far_clip + 0.01)) | planes(c2w))
        print(f"frame {k}: {N} surfels, {inv.sum()} in view ({inv.mean():.3f}); words with one in view {words.any(1).sum()} of {len(words)}, "
              f"lanes in view inside those {words[words.any(1)].mean():.2f}; tiles {len(need)}: needed {need.sum()}, round-1 box test {old.sum()}, "
              f"side planes for every box {new.sum()} (needed but culled: {(need & ~new).sum()}), with the rotated box as well {both.sum()} "
              f"(needed but culled: {(need & ~both).sum()})")
    o.process_frame(rgb, depth, sem, pose)
