#!/usr/bin/env python3
"""tools/pass_trace.py for BASELINE configs[2]: one k_surfel_pass launch over a 20 M-surfel model, workgroup by workgroup
(SM_PASS_TRACE: wall_clock64 at entry / first visited tile / after that tile's phase A / exit).  usage: tools/pass_trace_hd.py [wg_per_cu]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
prefix = os.path.join(ROOT, "gpurun_out", "pass_trace_hd")
os.environ["SM_PASS_TRACE"] = prefix
if len(sys.argv) > 1:
    os.environ["SM_PASS_WG_PER_CU"] = sys.argv[1]
from surfelmapping_amd import capi, synth   # noqa: E402
import bench                                 # noqa: E402

if __name__ == "__main__":
    cam = synth.HD
    frames = bench.make_frames(cam, 8, 1, 15.0, 8)
    P = cam["width"] * cam["height"]
    sm = capi.SurfelMap(capi.make_config(**cam, preprocess=0, conflict_cap=0, max_sqrt_vertices=10000))
    sm.upload_model(synth.seeded_model(20_000_000, tick=300, seed=1)); sm.set_tick(300)
    bufs = bench.stage_frames(sm, frames, P)
    for b in bufs:
        sm.process_frame_device(*b)
    sm.sync()
    print("counts", sm.counts())
    sm.close()
    t = np.fromfile(prefix + ".0.bin", dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    t0 = t[:, 0].min()
    ent, first, after, ex = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0, (t[:, 3] - t0) / 100.0
    ntiles = int(t[0, 7]); nwg = len(t)
    per = ntiles / nwg
    print(f"workgroups {nwg}, tiles {ntiles} ({per:.1f} per workgroup); launch span {ex.max():.1f} us")
    print("entry percentiles 0/50/90/100:", np.percentile(ent, [0, 50, 90, 100]).round(2))
    print("entry -> first tile 50/90:", np.percentile(first - ent, [50, 90]).round(2))
    print("first tile's phase A (loads, cheap tests, list append) 50/90/100:", np.percentile(after - first, [50, 90, 100]).round(2))
    print("first tile -> exit 50/90/100:", np.percentile(ex - first, [50, 90, 100]).round(2), " => per tile", (np.median(ex - first) / per).round(2), "us")
    print("exit percentiles 0/50/90/100:", np.percentile(ex, [0, 50, 90, 100]).round(2))
    late = ent > 5.0
    print(f"workgroups entering later than 5 us: {late.sum()}; their duration median {np.median((ex - ent)[late]) if late.any() else 0:.1f} vs {np.median((ex - ent)[~late]):.1f} us")
