#!/usr/bin/env python3
"""Per-workgroup time line of one k_assoc_prep<., true> launch (preprocess = 1: chain tiles | association | tile flags in dispatch
order), from SM_PASS_TRACE's (entry, exit) stamps.  usage: tools/chain_trace.py [frames]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
prefix = os.path.join(ROOT, "gpurun_out", "chain_trace")
os.environ["SM_PASS_TRACE"] = prefix
from surfelmapping_amd import capi, synth   # noqa: E402
import bench                                 # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    cam = synth.KITTI
    frames = bench.make_frames(cam, n, 1, 15.0, 8)
    P = cam["width"] * cam["height"]
    sm = capi.SurfelMap(capi.make_config(**cam, preprocess=1))
    bufs = bench.stage_frames(sm, frames, P)
    for b in bufs:
        sm.process_frame_device(*b)
    sm.sync()
    print("counts", sm.counts())
    sm.close()
    a = np.fromfile(prefix + ".assoc_prep.bin", dtype=np.uint64).astype(np.int64)
    n_assoc, n_flag, n_img = a[:3]
    n_fix, n_assoc = int(n_assoc >> 32), int(n_assoc & 0xFFFFFFFF)     # two-launch frame: publisher + repair crew open the grid
    a = a[3:].reshape(-1, 2)[n_fix:]
    a0 = a[:, 0].min()
    en, ex = (a[:, 0] - a0) / 100.0, (a[:, 1] - a0) / 100.0
    print(f"k_assoc_prep<chain>: {n_img} chain tiles + {n_assoc} association + {n_flag} flag workgroups, launch span {ex.max():.2f} us")
    for name, lo, hi in (("chain", 0, n_img), ("assoc", n_img, n_img + n_assoc), ("flags", n_img + n_assoc, n_img + n_assoc + n_flag)):
        if hi > lo:
            e, x = en[lo:hi], ex[lo:hi]
            print(f"  {name}: entry 0/50/100 {np.percentile(e, [0, 50, 100]).round(2)}  duration 50/90/100 {np.percentile(x - e, [50, 90, 100]).round(2)}  exit 50/90/100 {np.percentile(x, [50, 90, 100]).round(2)}")
