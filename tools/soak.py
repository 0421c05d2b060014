#!/usr/bin/env python3
"""One-off soak: a long synthetic drive (forward, turn, drive back) on the GPU against the CPU oracle, bit-exact
comparison every CHECK frames.  Not part of the test suite (minutes of CPU time)."""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol                      # noqa: E402
from surfelmapping_amd import capi, synth   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
CHECK = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cam = dict(width=320, height=120, fx=180.0, fy=180.0, cx=159.5, cy=59.5)
poses = []
z, yaw = 0.0, 0.0
for k in range(N):
    phase = (k // 150) % 4
    if phase in (0, 2):
        z += 0.8 * (1 if phase == 0 else -1)
    else:
        yaw += 180.0 / 150.0
    poses.append(synth.pose_matrix(0.3 * math.sin(k / 30.0), 0.0, z, yaw))
scene = synth.Scene(5, n_boxes=12, length=150.0)
over = dict(preprocess=0, stereo_border=20.0, max_sqrt_vertices=3000)
g = capi.SurfelMap(capi.make_config(**cam, **over))
o = ol.Oracle(ol.make_config(**cam, **over))
c = synth.Camera(**cam)
t0 = time.time()
for k, p in enumerate(poses):
    rgb, d, s = scene.render(c, p, noise_mm=3.0, noise_seed=k)
    pc = synth.pose_to_colmajor(p)
    g.process_frame(rgb, d, s, pc); o.process_frame(rgb, d, s, pc)
    if (k + 1) % CHECK == 0 or k == N - 1:
        a, b = g.download_model(), o.download_model()
        same = a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        lg = g.read_frame_log(1)[-1]
        print(f"frame {k + 1}: count {a.shape[0]} {'== oracle' if same else '!= ORACLE'}  conf_skipped {lg['n_conf_skipped']} "
              f"splat_skipped {lg['n_splat_skipped']} static {lg['n_static']}  {time.time() - t0:.0f}s", flush=True)
        if not same:
            sys.exit(1)
print("SOAK OK")
