#!/usr/bin/env python3
"""Frame-time diagnostic without torch / bench.py's machinery: 110 KITTI-size frames through the plain path or the in-stream
sharded path (one rank, no communicator), synchronising every CHUNK frames and printing the host enqueue time and the total
time per frame of each chunk.  (Used in round 2 to tell a GPU-side slowdown from a host-side stall: the 45 ms "stall" of
bench.py's sharded leg turned out to be Python's garbage collector with torch imported.)

    python tools/shard_diag.py {plain|shard} CHUNK
"""
import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from surfelmapping_amd import capi, sharded, synth
import bench
cam = dict(synth.KITTI)
n = 110
frames = bench.make_frames(cam, n, 7, 15.0, 8)
P = cam["width"] * cam["height"]
mode = sys.argv[1]
sm = capi.SurfelMap(capi.make_config(**cam, preprocess=0, conflict_cap=1))
if mode == "shard":
    mp = sharded.StreamShard(sm, 0, 1)
    step = sm.shard_frame_device
else:
    step = sm.process_frame_device
dptr = []
for rgb, depth, sem, pose in frames:
    dr, dd, ds = sm.device_alloc(P * 3), sm.device_alloc(P * 2), sm.device_alloc(P)
    sm.device_upload(dr, rgb); sm.device_upload(dd, depth); sm.device_upload(ds, sem)
    dptr.append((dr, dd, ds, pose))
chunk = int(sys.argv[2])
t_all = time.perf_counter()
for k0 in range(0, n, chunk):
    t0 = time.perf_counter()
    enq = []
    for k in range(k0, min(k0 + chunk, n)):
        te = time.perf_counter(); step(*dptr[k]); enq.append(time.perf_counter() - te)
    t1 = time.perf_counter()
    sm.sync()
    t2 = time.perf_counter()
    print(f"frames {k0:3d}.. enqueue {1e6*(t1-t0)/chunk:8.1f} us/frame (max {1e6*max(enq):9.1f})  total {1e6*(t2-t0)/chunk:8.1f} us/frame  count {sm.counts()['count']}", flush=True)
print("all", time.perf_counter() - t_all)
