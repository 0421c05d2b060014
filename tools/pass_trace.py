#!/usr/bin/env python3
"""One-off diagnosis: where the time of one k_surfel_pass launch goes, workgroup by workgroup.

SM_PASS_TRACE=<prefix> makes the HIP core hand the kernel a buffer of 8 words per workgroup (wall_clock64 -- 100 MHz -- at
entry, at its first visited tile, after that tile, at exit; the tile, its compacted entries, XCC | HW_ID, tiles) and dump
the last launch's record at sm_destroy.  This script runs N KITTI-shaped frames (bench.py's generator), dumps, and
prints the launch's time line.  usage: tools/pass_trace.py [frames] [compact 0|1] | --analyze"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
prefix = os.path.join(ROOT, "gpurun_out", "pass_trace")
if len(sys.argv) > 1 and sys.argv[1] == "--analyze":      # only print the time line of the dumps already in gpurun_out/
    pass
else:
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 23
    os.environ["SM_PASS_TRACE"] = prefix
    if len(sys.argv) > 2:
        pass
    from surfelmapping_amd import capi, synth   # noqa: E402
    import bench                                 # noqa: E402

    cam = synth.KITTI
    frames = bench.make_frames(cam, n_frames, 1, 15.0, 8)
    P = cam["width"] * cam["height"]
    sm = capi.SurfelMap(capi.make_config(**cam, preprocess=0))
    bufs = []
    for rgb, d, s, p in frames:
        dr, dd, ds = sm.device_alloc(P * 3), sm.device_alloc(P * 2), sm.device_alloc(P)
        sm.device_upload(dr, rgb); sm.device_upload(dd, d); sm.device_upload(ds, s)
        bufs.append((dr, dd, ds, p))
    for b in bufs:
        sm.process_frame_device(*b)
    sm.sync()
    print("counts", sm.counts())
    sm.close()
t = np.fromfile(prefix + ".0.bin", dtype=np.uint64).reshape(-1, 8).astype(np.int64)
t0 = t[:, 0].min()
ent, first, after, ex = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0, (t[:, 3] - t0) / 100.0     # us
tile, nact, xcc = t[:, 4], t[:, 5], t[:, 6] >> 32
visited = t[:, 4] != -1
print(f"workgroups {len(t)}, tiles {t[0, 7]}, visiting a tile {visited.sum()}, with entries {(nact > 0).sum()}; launch span {ex.max():.2f} us")
print("entry  (us after the first workgroup) percentiles 0/50/90/100:", np.percentile(ent, [0, 50, 90, 100]).round(2))
print("exit   percentiles 0/50/90/99/100:", np.percentile(ex, [0, 50, 90, 99, 100]).round(2))
v = visited
print("visited: entry -> first tile (DevState, flags)  50/90/100:", np.percentile((first - ent)[v], [50, 90, 100]).round(2))
print("visited: first tile duration                    50/90/100:", np.percentile((after - first)[v], [50, 90, 100]).round(2))
print("visited: after first tile -> exit               50/90/100:", np.percentile((ex - after)[v], [50, 90, 100]).round(2))
for lo, hi in ((0, 1), (1, 128), (128, 384), (384, 640), (640, 1025)):
    m = v & (nact >= lo) & (nact < hi)
    if m.any():
        print(f"entries [{lo},{hi}): {m.sum():5d} workgroups, tile time median {np.median((after - first)[m]):.2f} max {(after - first)[m].max():.2f} us, exit median {np.median(ex[m]):.2f} max {ex[m].max():.2f}")
order = np.argsort(-ex)[:12]
print("the last workgroups to leave: (block, xcc, tile, entries, entry, first, after, exit)")
for b in order:
    print(f"  {b:5d} {xcc[b]} {tile[b]:6d} {nact[b]:5d} {ent[b]:6.2f} {first[b]:6.2f} {after[b]:6.2f} {ex[b]:6.2f}")
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"xcc {x}: {m.sum()} workgroups, entries {nact[m].sum()}, last exit {ex[m].max():.2f}")
# ---- the last k_assoc_prep launch: (entry, exit) per workgroup; block ranges: association | tile flags | image tiles
ap = prefix + ".assoc_prep.bin"
if os.path.exists(ap):
    a = np.fromfile(ap, dtype=np.uint64).astype(np.int64)
    n_assoc, n_flag, n_img = a[:3]          # block ranges in dispatch order: association | tile flags | image tiles
    n_fix, n_assoc = int(n_assoc >> 32), int(n_assoc & 0xFFFFFFFF)     # two-launch frame: publisher + repair crew open the grid
    a = a[3:].reshape(-1, 2)
    if n_fix:
        f0 = a[:, 0].min()
        print(f"k_assoc_prep: {n_fix} fixup workgroups first: publisher {((a[0, 0] - f0) / 100.0):.2f} -> {((a[0, 1] - f0) / 100.0):.2f} us, crew exit max {((a[1:n_fix, 1] - f0).max() / 100.0):.2f}")
    a = a[n_fix:]
    a0 = a[:, 0].min()
    en, exi = (a[:, 0] - a0) / 100.0, (a[:, 1] - a0) / 100.0
    print(f"k_assoc_prep: {n_img} image + {n_flag} flag + {n_assoc} association workgroups, launch span {exi.max():.2f} us")
    for name, lo, hi in (("assoc", 0, n_assoc), ("flags", n_assoc, n_assoc + n_flag), ("image", n_assoc + n_flag, n_assoc + n_flag + n_img)):
        e, x = en[lo:hi], exi[lo:hi]
        print(f"  {name}: entry 0/50/100 {np.percentile(e, [0, 50, 100]).round(2)}  duration 50/90/100 {np.percentile(x - e, [50, 90, 100]).round(2)}  exit 50/90/100 {np.percentile(x, [50, 90, 100]).round(2)}")
    last = np.argsort(-exi)[:8]
    print("  last to leave (block, entry, exit):", [(int(b), round(float(en[b]), 2), round(float(exi[b]), 2)) for b in last])
