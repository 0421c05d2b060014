#!/usr/bin/env python3
"""Where the time of sm_process_frame_async goes: host time per call and frames/s with pageable caller buffers and with buffers of sm_host_alloc (KITTI size, preprocess = 1)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from surfelmapping_amd import capi, synth
import bench

if __name__ == "__main__":
    cam = synth.KITTI
    n = 45
    frames = bench.make_frames(cam, n, 1, 15.0, 8)
    for mode in ("pageable", "hostalloc", "device"):
        sm = capi.SurfelMap(capi.make_config(**cam, preprocess=1))
        if mode == "hostalloc":
            fr2 = []
            for rgb, d, s, p in frames:
                a, b, c = sm.host_frame()
                a[...] = rgb.reshape(a.shape); b[...] = d.reshape(b.shape); c[...] = s.reshape(c.shape)
                fr2.append((a, b, c, p))
            frames_use = fr2
        else:
            frames_use = frames
        dp = bench.stage_frames(sm, frames, cam["width"] * cam["height"]) if mode == "device" else None
        def call(k):
            if dp:
                return sm.process_frame_device(*dp[k])
            return sm.process_frame_async(*frames_use[k])
        for k in range(5):
            call(k)
        sm.sync()
        calls = []
        t0 = time.perf_counter()
        for k in range(5, n):
            c0 = time.perf_counter()
            call(k)
            calls.append(time.perf_counter() - c0)
        t_enq = time.perf_counter() - t0
        sm.sync()
        el = time.perf_counter() - t0
        print(mode, "frames/s", round((n - 5) / el), "us/frame", round(el / (n - 5) * 1e6, 1), "| host per call median", round(np.median(calls) * 1e6, 1), "max", round(max(calls) * 1e6, 1),
              "| enqueue total", round(t_enq * 1e6 / (n - 5), 1))
        sm.close()
    # the box's own H2D rate for one frame's worth of bytes (pinned and pageable), for reading the figures above
    try:
        import torch
        nbytes = cam["width"] * cam["height"] * 6
        dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        for kind in ("pinned", "pageable"):
            src = torch.empty(nbytes, dtype=torch.uint8).pin_memory() if kind == "pinned" else torch.empty(nbytes, dtype=torch.uint8)
            for _ in range(3):
                dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            print(f"H2D {kind}: {nbytes / 1e6:.1f} MB in {dt * 1e6:.0f} us = {nbytes / dt / 1e9:.1f} GB/s")
    except Exception as e:      # noqa: BLE001
        print("H2D probe skipped:", e)
