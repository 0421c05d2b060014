#!/usr/bin/env python3
"""One-off: aggregate throughput of K independent camera streams (K contexts, K HIP streams) on ONE GPU, frames
enqueued round-robin from one host thread.  The kernels of a single stream leave most of the chip idle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from surfelmapping_amd import capi, synth   # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, W0 = 100, 10
cam = synth.KITTI
frames = synth.make_sequence(cam, synth.kitti_trajectory(N + W0), seed=1, noise_mm=15.0)
P = cam["width"] * cam["height"]
ctxs = []
for k in range(K):
    sm = capi.SurfelMap(capi.make_config(**cam, preprocess=0))
    bufs = []
    for rgb, d, s, p in frames:
        dr, dd, ds = sm.device_alloc(P * 3), sm.device_alloc(P * 2), sm.device_alloc(P)
        sm.device_upload(dr, rgb); sm.device_upload(dd, d); sm.device_upload(ds, s)
        bufs.append((dr, dd, ds, p))
    ctxs.append((sm, bufs))
for f in range(W0):
    for sm, bufs in ctxs:
        sm.process_frame_device(*bufs[f])
for sm, _ in ctxs:
    sm.sync()
t0 = time.perf_counter()
for f in range(W0, W0 + N):
    for sm, bufs in ctxs:
        sm.process_frame_device(*bufs[f])
for sm, _ in ctxs:
    sm.sync()
el = time.perf_counter() - t0
print(f"{K} streams: {K * N / el:.0f} frames/s aggregate, {el / N * 1e6:.1f} us per round of {K} frames, counts {[c[0].counts()['count'] for c in ctxs]}")
