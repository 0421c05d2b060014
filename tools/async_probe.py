#!/usr/bin/env python3
"""Where the time of sm_process_frame_async goes: host time per call and frames/s with registered / pageable / hipHostMalloc'd
caller buffers (KITTI size, preprocess = 1)."""
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
    for mode in ("pageable", "registered", "hostalloc", "ring3", "device"):
        sm = capi.SurfelMap(capi.make_config(**cam, preprocess=1))
        if mode == "registered":
            t0 = time.perf_counter()
            for rgb, d, s, _ in frames:
                sm.pin_host(rgb); sm.pin_host(d); sm.pin_host(s)
            print("  hipHostRegister of", 3 * n, "arrays:", round((time.perf_counter() - t0) * 1e3, 1), "ms")
        if mode == "hostalloc":
            fr2 = []
            for rgb, d, s, p in frames:
                a, b, c = sm.host_array(rgb.shape, rgb.dtype), sm.host_array(d.shape, d.dtype), sm.host_array(s.shape, s.dtype)
                a[...] = rgb; b[...] = d; c[...] = s
                fr2.append((a, b, c, p))
            frames_use = fr2
        elif mode == "ring3":            # a reader that reuses three registered buffer sets (its decode = a copy here, in the loop)
            ring = [tuple(np.empty_like(x) for x in frames[0][:3]) for _ in range(3)]
            for r in ring:
                for x in r:
                    sm.pin_host(x)
            frames_use = None
        else:
            frames_use = frames
        dp = bench.stage_frames(sm, frames, cam["width"] * cam["height"]) if mode == "device" else None
        def call(k):
            if dp:
                return sm.process_frame_device(*dp[k])
            if frames_use is None:
                r = ring[k % 3]
                if k >= 3:
                    sm.inputs_consumed()
                for dst, src in zip(r, frames[k][:3]):
                    np.copyto(dst, src)
                return sm.process_frame_async(r[0], r[1], r[2], frames[k][3])
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
