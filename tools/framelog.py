#!/usr/bin/env python3
"""Print the device-side frame log of a short KITTI-shaped run (counters per frame): how many conflicts a frame has
relative to W*H (the conflict cap), how many surfels are in view, fused, new."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from surfelmapping_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
ft = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
cam = synth.KITTI
seq = synth.make_sequence(cam, synth.kitti_trajectory(n), seed=1, noise_mm=noise)
sm = capi.SurfelMap(capi.make_config(**cam, preprocess=0, fuse_thresh=ft))
for fr in seq:
    sm.process_frame(*fr)
log = sm.read_frame_log(n)
P = cam["width"] * cam["height"]
print("P =", P)
for e in log:
    print({k: int(e[k]) for k in log.dtype.names})
