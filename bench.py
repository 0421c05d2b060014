#!/usr/bin/env python3
"""bench.py -- frames/sec of the per-frame surfel-fusion hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one SurfelMapping::processFrame of the hot path (metricise -> conflict/cull ->
index-map splat -> associate/fuse -> append) over one 1242x375 KITTI-shaped synthetic
RGB-D+semantic frame (BASELINE.json configs[1]); frames are resident in HBM before the timed
region starts and are enqueued back-to-back (no host read-back inside a frame or between
frames -- the counters of every frame are audited afterwards from the device-side frame log).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: N fresh
child processes of this file (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one
GPU each) -- the parent never touches a GPU -- and relays rank 0's one line.  Under torchrun
(WORLD_SIZE set) the process is one of the ranks.  WORLD_SIZE != --gpus is an error, never a
silent one-GPU run.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against the 8 TB/s HBM
peak with algorithmic bytes from the per-frame counters (DESIGN.md "Measurement");
`cpu_baseline` is the CPU oracle (oracle/, a scalar restatement of the reference's passes)
timed on this host over the same frames, 1 thread.  One rank also reports, in the same line:
`steady_leg` (100 frames after 10), `fuse_leg` (frames that actually fuse), `reference_path_leg`
(preprocess = 1, host buffers: what the drop-in's SurfelMapping::processFrame executes) and
`hd_leg` (BASELINE configs[2]: 1920x1080, 20 M seeded surfels).  N ranks report the rig
aggregate as `value` (configs[4]: one camera per GPU), the consolidation into a single
GlobalModel, and `sharded_leg` (configs[3]: ONE stream over the N GPUs).
"""
from __future__ import annotations

import argparse
import gc
import json
import math
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from surfelmapping_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "frames/sec per GPU, 1242x375 KITTI-shaped RGB-D+semantic, full associate+fuse+merge"


# ------------------------------------------------------------------------------------------------
# synthetic frames (CPU only; forked workers never touch a GPU)
# ------------------------------------------------------------------------------------------------

def ordinal(n):
    """16 -> '16th', 32 -> '32nd'"""
    return f"{n}{'th' if 10 <= n % 100 <= 20 else {1: 'st', 2: 'nd', 3: 'rd'}.get(n % 10, 'th')}"

def rig_trajectory(n, rank, world, step=0.8):
    """Camera `rank` of a `world`-camera rig: same forward motion, yaw offset rank*360/world deg
    (BASELINE configs[4] / SURVEY 8d config 5); world == 1 is the plain KITTI trajectory."""
    yaw0 = 360.0 / world * rank if world > 1 else 0.0
    return [synth.pose_matrix(0.0, 0.0, step * k, yaw0 + 0.5 * math.sin(k / 20.0)) for k in range(n)]


def _render(args):
    cam_kw, pose, seed, k, noise = args
    scene = synth.Scene(seed)
    rgb, depth, sem = scene.render(synth.Camera(**cam_kw), pose, noise_mm=noise, noise_seed=seed * 100003 + k)
    return rgb, depth, sem, synth.pose_to_colmajor(pose)


def make_frames(cam_kw, n, seed, noise, workers, rank=0, world=1):
    poses = rig_trajectory(n, rank, world)
    jobs = [(cam_kw, poses[k], seed, k, noise) for k in range(n)]
    if workers > 1:
        with mp.get_context("fork").Pool(workers) as pool:
            return pool.map(_render, jobs, chunksize=max(1, n // (workers * 4)))
    return [_render(j) for j in jobs]


# ------------------------------------------------------------------------------------------------
# N ranks from one command line
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`--gpus N` outside torchrun: N fresh child processes, one per GPU.  The parent has not imported torch or the HIP core
    and never will; it relays rank 0's stdout (the one JSON line), lets every rank's stderr through, and fails if any rank does."""
    n = args.gpus
    env0 = dict(os.environ)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env0["MASTER_ADDR"] = env0.get("MASTER_ADDR", "127.0.0.1")
    env0["MASTER_PORT"] = str(_free_port())
    env0["WORLD_SIZE"] = str(n)
    env0["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE))
    # wait for all of them; the first rank that fails (or the time limit) ends the others -- the exact PIDs started above --
    # so that nobody is left waiting inside a collective
    deadline = time.time() + args.launch_timeout
    failed = False
    while any(pr.poll() is None for pr in procs):
        if any(pr.poll() not in (None, 0) for pr in procs) or time.time() > deadline:
            failed = True
            for q in procs:
                if q.poll() is None:
                    q.kill()
            break
        time.sleep(0.05)
    outs = [pr.communicate()[0].decode(errors="replace") for pr in procs]       # (one short line per rank: far below the pipe buffer)
    rcs = [pr.returncode for pr in procs]
    if failed or any(rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs}" + (" (time limit)" if time.time() > deadline else "") + "\n")
        raise SystemExit(1)
    if args.launch_dry_run:
        kids = sorted((json.loads(o.strip().splitlines()[-1]) for o in outs), key=lambda d: d["rank"])
        print(json.dumps({"launch_dry_run": True, "n_gpus": n, "ranks": [k["rank"] for k in kids], "children": kids}))
        return
    # rank 0's JSON line is the run's one line of stdout; anything else a rank (or a library inside it) wrote to stdout goes to
    # stderr, so that a banner cannot cost the measurement
    result = None
    for r, out in enumerate(outs):
        for ln in out.splitlines():
            if not ln.strip():
                continue
            if r == 0 and ln.lstrip().startswith("{") and '"metric"' in ln:
                result = ln
            else:
                sys.stderr.write(f"[rank {r} stdout] {ln}\n")
    if result is None:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        raise SystemExit(1)
    print(result)

def dry_run_rank(rank, world, local_rank):
    """--launch-dry-run: what a child would bind to, plus a gloo all-reduce over the ranks (proves the rendezvous the launcher
    set up); no GPU is touched, so this runs in the CPU test suite."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    dist.destroy_process_group()
    print(json.dumps({"rank": rank, "world": world, "local_rank": local_rank, "master": f"{os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']}",
                      "sum_of_ranks_plus_1": int(t[0]), "pid": os.getpid()}))


# ------------------------------------------------------------------------------------------------
# roofline bookkeeping
# ------------------------------------------------------------------------------------------------
def kernel_bytes(log):
    """Algorithmic HBM bytes per launch of each kernel from the per-frame counters
    (DESIGN.md "Measurement"; SoA surfel = 44 B, key = 8 B).  Every entry is an array over the timed frames; a kernel
    that runs only on some frames is averaged over those frames by the caller."""
    P = log["P"]
    N, Np, V, F, U, Ns, Cs, Ss, Sl = (log[k].astype(np.float64) for k in
                                      ("n_before", "n_after_cull", "visible_count", "fused_count", "unstable_count", "n_static",
                                       "n_conf_skipped", "n_splat_skipped", "n_slots"))
    # Sl = slots the cull scans (live surfels + slots of surfels killed since the last physical compaction).
    # Slots that stay in place (n_static; on a deferred-compaction frame that is all of them) are only read for the
    # splat (pos_conf 16 + time 4) unless their tile's bounding box is out of view (then not at all); on a compacting
    # frame the rest is read in full (44) and its survivors rewritten (44); every drawn surfel costs one 8-byte key atomic
    compact = 20.0 * np.maximum(Ns - Ss, 0.0) + 44.0 * np.maximum(Sl - Ns, 0.0) + 44.0 * np.maximum(Np - Ns, 0.0) + 8.0 * V
    # one-pass frames: pos_conf (16) of every slot whose tile is not skipped by BOTH tests (>= Sl - min(Cs, Ss): the
    # exact count is not logged, this is the lower bound), the time (4) of the slots that reach the exact tests (not
    # logged; lower bound: the drawn ones), one key atomic per drawn surfel
    one_pass = 16.0 * np.maximum(Sl - np.minimum(Cs, Ss), 0.0) + 4.0 * V + 8.0 * V
    pre = 6.0 * P + 24.0 * P                                     # u8x3+u16+u8 in, f32+u32+u64+(f32,u32) out
    return {
        "k_prep": np.full_like(N, pre),
        "k_conflict": 16.0 * np.maximum(Sl - Cs, 0.0),            # tiles skipped by their bounds are not read
        "k_compact": compact,
        "k_surfel_pass": one_pass,
        "k_pass_fixup": np.full_like(N, 4.0 * P),                 # the candidate count reads the depth plane once
        "k_associate": 16.0 * P + 84.0 * F,                       # depth+rgbs+key per pixel, gather 44 + scatter 40 per fuse
        "k_append": 8.0 * (P / 64.0) + 44.0 * U,
        "k_associate_direct": 16.0 * P + 84.0 * F + 44.0 * U,     # ... and the new surfels written in place
    }


def kernel_table(log, tim, P, K, warmup, compact_period, workload):
    """Per-kernel launches, average duration (HIP events), algorithmic bytes and GB/s of the timed frames, and the
    roofline entry of the kernel the run spends most time in.  The frame forms: a frame whose cull only
    marks the dead runs k_surfel_pass + k_pass_fixup (+ k_associate_direct when it appends directly), the others
    k_conflict + k_scan_cull/k_cull_finalize + k_compact (or k_cull_lazy) and k_associate + k_append_scan."""
    logd = {k: log[k] for k in log.dtype.names}
    logd["P"] = P
    kb = kernel_bytes(logd)
    n_op, n_dir, n_comp = int(tim.get("frames_one_pass", 0)), int(tim.get("frames_direct", 0)), int(tim.get("frames_compact", 0))
    Kt = min(K, int(tim.get("frames", K)) or K)      # the event ring holds the last 256 frames: launch counts are of those
    moved = log["n_static"] < log["n_slots"] if len(log) else np.zeros(0, bool)      # frames that compacted
    # (one-pass <=> not compacted on the default path; direct <=> one-pass and k_prep evaluated the tile flags)
    sel_op = ~moved if n_op else np.zeros(len(log), bool)
    sel_dir = sel_op if n_dir == n_op else np.zeros(len(log), bool)
    # asynchronous streams: a frame's association is held back and runs in the NEXT frame's preparation launch (k_assoc_prep)
    n_mrg, n_alone = int(tim.get("frames_merged", 0)), int(tim.get("frames_assoc_alone", n_dir))
    sel_mrg = np.zeros(len(log), bool)
    if n_mrg and len(log):
        sel_mrg[1:] = sel_dir[:-1]                        # frame k's launch carries frame k-1's association
        sel_mrg[0] = n_mrg > int(sel_mrg.sum())           # (the last warm-up frame's, if the counts say so)
        kb["k_assoc_prep"] = kb["k_prep"] + np.concatenate([kb["k_associate_direct"][:1], kb["k_associate_direct"][:-1]])
    rows = [("k_prep", tim.get("k_prep_own", tim["k_prep"]) if n_mrg else tim["k_prep"], ~sel_mrg, Kt - n_mrg),
            ("k_assoc_prep", tim.get("k_assoc_prep", 0.0), sel_mrg, n_mrg),
            ("k_surfel_pass", tim.get("k_surfel_pass", 0.0), sel_op, n_op),
            ("k_pass_fixup", tim.get("k_pass_fixup", 0.0), sel_op, n_op),
            ("k_associate_direct", tim.get("k_associate_direct", 0.0), sel_dir, n_alone),
            ("k_conflict", tim.get("k_conflict_own", 0.0), ~sel_op, Kt - n_op),
            ("k_compact", tim.get("k_compact_own", 0.0), moved, n_comp),
            ("k_associate", tim.get("k_associate_own", 0.0), ~sel_dir, Kt - n_dir),
            ("k_append", tim.get("k_append_own", 0.0), ~sel_dir, Kt - n_dir)]
    if Kt - n_op - n_comp > 0:      # lazy culls outside the one-pass form
        rows.append(("k_cull_lazy", tim.get("k_cull_lazy", 0.0), ~moved & ~sel_op, Kt - n_op - n_comp))
    kern, launches = {}, {}
    for name, ms, sel, n in rows:
        if n <= 0:
            continue
        b = kb["k_compact" if name == "k_cull_lazy" else name]
        mb = float(b[sel].mean()) / 1e6 if len(log) and sel.any() else 0.0
        kern[name] = {"ms": ms, "MB": mb, "GBs": (mb / 1e3) / (ms * 1e-3) if ms > 0 else None, "launches": n}
        launches[name] = n
    if Kt - n_op > 0:
        kern["k_scan_cull+k_cull_finalize"] = {"ms": tim.get("k_scan_own", 0.0), "MB": None, "GBs": None, "launches": Kt - n_op}
    # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very command
    # (profiles/README.md), gfx950-corrected per kernel by tools/prof_summary.py
    tj = None
    for tpath in (os.path.join(ROOT, "profiles", f"traffic_{workload}_s{K}_w{warmup}.json"), os.path.join(ROOT, "profiles", f"traffic_{workload}.json")):
        if os.path.exists(tpath):
            cand = json.load(open(tpath))
            if cand.get("steps") == K and cand.get("warmup") == warmup and cand.get("compact_period", 24) == compact_period:
                tj = cand
                break
    for name in kern:
        t = (tj or {}).get("kernels", {}).get(name, {})
        kern[name]["hbm_traffic_MB"] = t.get("hbm_bytes_per_launch") / 1e6 if t.get("hbm_bytes_per_launch") else None
        kern[name]["frac_of_hbm_peak"] = (kern[name]["GBs"] / HBM_PEAK_GBS) if kern[name].get("GBs") else None
    # dominant kernel = the one the run spends most time in (average duration x launches)
    dom = max(launches.keys(), key=lambda n: kern[n]["ms"] * launches[n])
    achieved = kern[dom]["GBs"] or 0.0
    t = (tj or {}).get("kernels", {}).get(dom, {})
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": t.get("hbm_bytes_per_launch"),
                "ms_per_launch": kern[dom]["ms"], "alg_bytes_per_launch": kern[dom]["MB"] * 1e6,
                "launches": launches[dom], "valu_issue_util": t.get("valu_issue_util"),
                "traffic_source": (os.path.relpath(tpath, ROOT) if tj else None)}
    # for information: the same kernel priced with the bytes the counters saw instead of the algorithmic ones (`frac` stays the
    # contract's figure; this one says how close the launch runs to the memory system with everything it actually moves)
    if roofline["traffic"] and kern[dom]["ms"] > 0:
        roofline["traffic_frac"] = roofline["traffic"] / (kern[dom]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    return kern, launches, roofline


# ------------------------------------------------------------------------------------------------
# one single-GPU leg
# ------------------------------------------------------------------------------------------------
def stage_frames(sm, frames, P):
    out = []
    for rgb, depth, sem, pose in frames:
        dr, dd, ds = sm.device_alloc(P * 3), sm.device_alloc(P * 2), sm.device_alloc(P)
        sm.device_upload(dr, rgb); sm.device_upload(dd, depth); sm.device_upload(ds, sem)
        out.append((dr, dd, ds, pose))
    return out


def run_leg(capi, cam, frames, K, Wm, cfg_kw, *, events=True, mode="device", seed_model=None, seed_tick=None, keep=False):
    """K frames after Wm on an un-instrumented context (this is the leg's frames/s), then -- events=True -- the same frames
    again on a context that records a HIP event after every kernel (per-kernel durations for the roofline).
    mode: "device"     frames staged in HBM, sm_process_frame_device, no host wait inside the K frames (the metric's form);
          "host_sync"  sm_process_frame on host buffers: 2.8 MB H2D + a host wait per frame = the reference's processFrame;
          "host_async" sm_process_frame_async on host buffers: pinned staging, the copy of frame f+1 overlaps frame f;
          "device_sync" device-resident frames, host waits for the counters after every frame.
    Returns a dict; with keep=True the timed context stays open under "sm"."""
    P = cam["width"] * cam["height"]

    def prime(sm):
        if seed_model is not None:
            sm.upload_model(seed_model)
            sm.set_tick(seed_tick)

    def step_fn(sm, dptr):
        if mode == "device":
            return lambda k: sm.process_frame_device(*dptr[k])
        if mode == "device_sync":
            def f(k):
                sm.process_frame_device(*dptr[k]); sm.sync()
            return f
        if mode == "host_sync":
            return lambda k: sm.process_frame(*frames[k])
        if mode == "host_async":
            return lambda k: sm.process_frame_async(*frames[k])
        raise ValueError(mode)

    sm = capi.SurfelMap(capi.make_config(**cam, **cfg_kw, enable_timing=0))
    prime(sm)
    dptr = stage_frames(sm, frames, P) if mode.startswith("device") else None
    if mode == "host_async":            # a reader that decodes into pinned frame blocks of the library (sm_host_alloc_frame); here one per frame
        pinned = []
        for rgb, depth, sem, pose in frames:
            a, b, c = sm.host_frame()
            a[...] = rgb.reshape(a.shape); b[...] = depth.reshape(b.shape); c[...] = sem.reshape(c.shape)
            pinned.append((a, b, c, pose))
        frames = pinned
    step = step_fn(sm, dptr)
    # no collector pause inside the timed region: with torch imported a full collection takes ~45 ms.  Collect BEFORE the
    # warm-up: a pause between warm-up and t0 lets the GPU clock down (+130 us on the first frames)
    gc.collect(); gc.disable()
    for k in range(Wm):
        step(k)
    sm.sync()
    t0 = time.perf_counter()
    for k in range(Wm, Wm + K):
        step(k)
    t_enq = time.perf_counter() - t0
    sm.sync()
    el = time.perf_counter() - t0
    gc.enable()
    res = {"fps": K / el, "ms": el / K * 1e3, "log": sm.read_frame_log(K), "counts": sm.counts(), "tim": None,
           "host_enqueue_ms": t_enq / K * 1e3}
    if keep:
        res["sm"], res["dptr"] = sm, dptr
    else:
        sm.close()
    if events:
        ev = capi.SurfelMap(capi.make_config(**cam, **cfg_kw, enable_timing=1))
        prime(ev)
        dp = stage_frames(ev, frames, P)
        for k in range(Wm):
            ev.process_frame_device(*dp[k])
        ev.sync(); ev.timings()                       # drop the warm-up samples
        for k in range(Wm, Wm + K):
            ev.process_frame_device(*dp[k])
        ev.sync()
        res["tim"] = ev.timings()
        assert ev.counts()["count"] == res["counts"]["count"], "instrumented pass diverged from the timed pass"
        ev.close()
    return res


def leg_summary(res, K):
    log = res["log"]
    return {"value": res["fps"], "unit": "frames/s", "ms_per_step": res["ms"],
            "fused_F_per_frame": float(log["fused_count"].mean()) if len(log) else 0.0,
            "new_U_per_frame": float(log["unstable_count"].mean()) if len(log) else 0.0,
            "conflicts_per_frame": float(log["conflict_count"].mean()) if len(log) else 0.0,
            "frames_that_compacted": int((log["n_static"] < log["n_slots"]).sum()) if len(log) else 0,
            "surfels_start": int(log["n_before"][0]) if len(log) else 0, "surfels_end": int(res["counts"]["count"])}


def oracle_run(ol, cam, frames, Wm, Kc, cfg_kw, libpath=None, seed_model=None, seed_tick=None, per_frame=False):
    """the CPU oracle over frames [0, Wm + Kc); returns (frames/s over the last Kc, final counts, per-frame counts)"""
    o = ol.Oracle(ol.make_config(**cam, **cfg_kw), **({"libpath": libpath} if libpath else {}))
    if seed_model is not None:
        o.upload_model(seed_model); o.set_tick(seed_tick)
    for k in range(Wm):
        o.process_frame(*frames[k])
    seq = []
    o.stage_seconds(reset=True)
    c0 = time.perf_counter()
    for k in range(Wm, Wm + Kc):
        o.process_frame(*frames[k])
        if per_frame:
            seq.append(o.counts())
    el = time.perf_counter() - c0
    oc = o.counts()
    oc["_ms_per_frame_by_pass"] = {k: v / Kc * 1e3 for k, v in o.stage_seconds().items()}     # where the CPU's time goes
    o.close()
    return Kc / el, oc, seq


# ------------------------------------------------------------------------------------------------
# N ranks: the rig (configs[4], `value`), its consolidation, and ONE stream sharded over the ranks (configs[3])
# ------------------------------------------------------------------------------------------------
def bench_ranks(args, cam, rank, world, local_rank, K, Wm, workers, emit):
    P = cam["width"] * cam["height"]
    t0 = time.time()
    frames = make_frames(cam, Wm + K, args.seed, args.noise_mm, workers, rank, world)           # this rank's camera
    shared = frames if rank == 0 else make_frames(cam, Wm + K, args.seed, args.noise_mm, workers, 0, 1)   # rank 0's stream, on every rank
    t_gen = time.time() - t0
    # torch first: its HIP runtime must be the one the core binds to (one runtime per process)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world == 1:                         # --force-dist outside a launcher: a one-rank rendezvous of its own
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
    # --rehearse-one-gpu: every rank on GPU 0, torch.distributed over gloo, the core's collectives staged through the host
    # (sharded.GlooCollective) -- the whole multi-process flow where RCCL cannot run (it refuses two ranks on one device)
    reh = args.rehearse_one_gpu
    if reh:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if reh:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    tdev = torch.device("cpu") if reh else dev     # where the handful of scalars torch.distributed reduces live
    from surfelmapping_amd import capi
    from surfelmapping_amd import dist as smd
    from surfelmapping_amd import sharded as smsh
    if not reh:
        os.environ.setdefault("SM_COMPACT_TICKETS", "0")     # this process's contexts never run at the same time (rehearsal: the ranks share the GPU, the core finds out by itself)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def bcast_id():
        ids = [capi.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        return ids[0]

    cfg = dict(preprocess=args.preprocess, device=local_rank, conflict_cap=1, compact_period=args.compact_period)
    msv_global = int(math.ceil(math.sqrt(world) * 5000))
    # ---- rig leg: one camera stream per GPU, no collective inside a frame
    sm = capi.SurfelMap(capi.make_config(**cam, **cfg))
    dptr = stage_frames(sm, frames, P)
    # the single GlobalModel DURING the run (sm_rig_consolidate_step: new-surfel lists all-gathered, the union cleaned against every
    # camera's latest view): one step after the warm-up, one after the timed frames -- the second is timed, outside `value`
    sm_inc = capi.SurfelMap(capi.make_config(**cam, preprocess=0, device=local_rank, max_sqrt_vertices=msv_global))
    rig = smd.RigMapper(sm, smsh.TorchComm(device_index=local_rank), P)
    # The core binds RCCL itself; should that fail on any rank (all ranks decide together), the collectives fall back to torch's own
    # group, staged through the host: `value` has no collective in it, only the consolidations and the sharded leg get slower
    rccl_note = None

    def native_collective(ctx):
        nonlocal rccl_note
        if reh:
            return smsh.GlooCollective(ctx), None
        ok, err = 1, ""
        the_id = bcast_id()
        try:
            ctx.rig_configure(rank, world)
            if os.environ.get("SM_BENCH_FAIL_RCCL"):      # (test hook: take the fallback)
                raise RuntimeError("SM_BENCH_FAIL_RCCL")
            ctx.shard_rccl_init(the_id)
        except Exception as e:      # noqa: BLE001
            ok, err = 0, repr(e)
        flag = torch.tensor([ok], dtype=torch.int64, device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 1:
            return "rccl-bound", None
        if ok:
            ctx.shard_rccl_finalize()
        rccl_note = f"RCCL binding failed on a rank ({err or 'another rank'}): collectives through torch.distributed, host-staged"
        sys.stderr.write("bench.py: " + rccl_note + "\n")
        return smsh.TorchCollective(ctx, device=dev), None

    coll, _ = native_collective(sm)
    if coll == "rccl-bound":
        rig._native = True                 # (configured and bound above)
        nranks_rig = sm.shard_rccl_nranks()
    else:
        rig.enable_native(coll)
        nranks_rig = None
    gc.collect(); gc.disable()
    for k in range(Wm):
        sm.process_frame_device(*dptr[k])
    sm.sync()
    rig.last = (frames[Wm - 1][1], frames[Wm - 1][2], frames[Wm - 1][3])
    rig.consolidate_step_native(sm_inc)
    barrier()
    t0 = time.perf_counter()
    for k in range(Wm, Wm + K):
        sm.process_frame_device(*dptr[k])
    sm.sync()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0                # this rank's K frames are complete; MAX over the ranks below
    barrier()                                     # (the closing barrier's own latency, ~0.4 ms, is not K frames' work)
    gc.enable()
    elapsed = max_over_ranks(own)
    log = sm.read_frame_log(K)
    fu = torch.tensor([float(log["fused_count"].sum()), float(log["unstable_count"].sum())], dtype=torch.float64, device=tdev)
    dist.all_reduce(fu, op=dist.ReduceOp.SUM)
    F_total, U_total = float(fu[0]), float(fu[1])
    own_t = torch.tensor([own], dtype=torch.float64, device=tdev)
    dist.broadcast(own_t, src=0)
    plain_ms = float(own_t[0]) / K * 1e3           # rank 0's camera IS the shared stream: the plain single-GPU time of those frames
    # ---- consolidation into a single GlobalModel (sm_rig_consolidate: all-gathers + per-slice cleanPoints, inside the core)
    rig.last = (frames[Wm + K - 1][1], frames[Wm + K - 1][2], frames[Wm + K - 1][3])     # the camera's latest view
    barrier()
    s0 = time.perf_counter()
    step_new, step_total = rig.consolidate_step_native(sm_inc)
    torch.cuda.synchronize()
    step_ms = max_over_ranks((time.perf_counter() - s0) * 1e3)
    sm_inc.close()
    sm_global = capi.SurfelMap(capi.make_config(**cam, preprocess=0, device=local_rank, max_sqrt_vertices=msv_global))
    barrier()
    g0 = time.perf_counter()
    global_count, view_conflicts = rig.consolidate_native(sm_global)
    torch.cuda.synchronize()
    gather_ms = max_over_ranks((time.perf_counter() - g0) * 1e3)
    if not reh and coll == "rccl-bound":
        sm.shard_rccl_finalize()
    counts = sm.counts()
    sm_global.close()
    sm.close()
    # ---- sharded leg: ONE stream (rank 0's camera) split over the ranks, bit-identical to one GPU (sm_shard_frame_device)
    sharded_leg = None
    if not args.no_sharded_leg:
        ss = capi.SurfelMap(capi.make_config(**cam, **cfg))
        if reh or rccl_note:
            shard = smsh.StreamShard(ss, rank, world, smsh.TorchCollective(ss, device=None if reh else dev))
            nranks_sh = None
        else:
            shard = smsh.StreamShard(ss, rank, world, "rccl", bcast_id())
            nranks_sh = ss.shard_rccl_nranks()
        dsh = stage_frames(ss, shared, P)
        gc.collect(); gc.disable()
        for k in range(Wm):
            ss.shard_frame_device(*dsh[k])
        ss.sync()
        barrier()
        t0 = time.perf_counter()
        for k in range(Wm, Wm + K):
            ss.shard_frame_device(*dsh[k])
        ss.sync()
        torch.cuda.synchronize()
        s_own = time.perf_counter() - t0
        barrier()
        gc.enable()
        s_el = max_over_ranks(s_own)
        sc = ss.counts()
        chk = torch.tensor([sc["count"], sc["conflict_count"], sc["unstable_count"]], dtype=torch.int64, device=tdev)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if nranks_sh is not None:
            ss.shard_rccl_finalize()
        ss.close()
        del shard
        sharded_leg = {"config": f"BASELINE configs[3]: ONE KITTI 1242x375 stream sharded over {world} GPUs (slot-addressed shards; per frame an "
                                 "all-reduce(min) of the 3.7 MB key map and an all-reduce(sum) of the fused-pixel mask + 3 counters, RCCL called from the "
                                 "HIP core on the frame's stream; deferred compaction between frames), bit-identical to one GPU",
                       "value": K / s_el, "unit": "frames/s", "ms_per_step": s_el / K * 1e3, "scaling": "strong",
                       "plain_single_gpu_ms_per_step": plain_ms, "sharded_over_plain": (s_el / K * 1e3) / plain_ms,
                       "counters_identical_on_all_ranks": bool(torch.equal(lo, hi)), "surfels_end": int(sc["count"]),
                       "plain_surfels_end_rank0": None, "rccl_nranks": nranks_sh}
        c0 = torch.tensor([counts["count"]], dtype=torch.int64, device=tdev)
        dist.broadcast(c0, src=0)
        sharded_leg["plain_surfels_end_rank0"] = int(c0[0])
        sharded_leg["same_surfel_count_as_plain"] = int(c0[0]) == int(sc["count"])
    dist.destroy_process_group()
    if rank != 0:
        return
    emit({
        "metric": METRIC, "value": world * K / elapsed, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[4] at the metric's image size: {world}-camera rig (yaw k*{360.0 / world:.0f} deg), one KITTI 1242x375 "
                               f"stream per GPU, depth noise {args.noise_mm} mm, metricise+conflict+cull+splat+associate+fuse+append per frame; "
                               "no collective inside a frame",
                   "frames": f"{Wm}..{Wm + K - 1}",
                   "multi_gpu": {"consolidation": "after the timed frames: every slice cleaned against all latest views (cleanPoints per view, the W*H "
                                                  "conflict cap shared exactly across the slices), slices all-gathered in rank order into a single "
                                                  "GlobalModel on every rank, inside the HIP core over RCCL (sm_rig_consolidate); not in `value`",
                                 "consolidation_ms": gather_ms,
                                 "incremental": {"what": "sm_rig_consolidate_step once per K frames: the ranks' new-surfel lists all-gathered into the single "
                                                         "GlobalModel (kept on every rank), which is then cleaned against every camera's latest view",
                                                 "every_K_frames": K, "step_ms": step_ms, "ms_per_frame": step_ms / K,
                                                 "new_surfels_exchanged": int(step_new), "global_model_surfels": int(step_total)},
                                 "global_model_surfels": int(global_count), "conflicts_per_view": [int(c) for c in view_conflicts]},
                   "host_sync": "none inside the timed region", "surfels_end_rank0": int(counts["count"])},
        "rccl": ({"nranks": None, "world_size": world, "backend": "REHEARSAL on one GPU: torch.distributed gloo, the core's collectives staged through "
                                                                    "the host (sharded.GlooCollective); not a measurement of anything"} if reh else
                 {"nranks": nranks_rig, "world_size": world, "backend": rccl_note or "RCCL bound by the HIP core (ncclCommInitRank / ncclCommCount); "
                                                                          "torch.distributed only hands the communicator id round"}),
        "surfels_fused_per_sec": (F_total + U_total) / elapsed,
        "fused_F": {"total": F_total, "per_sec": F_total / elapsed, "per_frame": F_total / (K * world)},
        "new_U": {"total": U_total, "per_sec": U_total / elapsed, "per_frame": U_total / (K * world)},
        "sharded_leg": sharded_leg,
        "roofline": None, "cpu_baseline": None,       # N = 1 reports them (cpu_baseline: rank 0 at N = 1 only)
        "gen_seconds": t_gen})


# ------------------------------------------------------------------------------------------------
# one rank: the metric's line with its legs
# ------------------------------------------------------------------------------------------------
def bench_single(args, cam, K, Wm, workers, emit):
    hd = args.workload == "hd20m"
    P = cam["width"] * cam["height"]
    t0 = time.time()
    n_steady = 0 if (hd or args.no_steady_leg) else args.steady_warmup + args.steady_steps
    frames = make_frames(cam, max(Wm + K, n_steady), args.seed, args.noise_mm, workers)
    t_gen = time.time() - t0
    from surfelmapping_amd import capi
    # this process creates several contexts one after the other; they never run at the same time: keep the default
    # round-robin form of the compaction kernel (the library would switch to ticket order for contexts that share a GPU)
    os.environ.setdefault("SM_COMPACT_TICKETS", "0")
    base = dict(preprocess=args.preprocess, conflict_cap=0 if hd else 1, max_sqrt_vertices=10000 if hd else 5000,
                compact_period=args.compact_period)
    seed_model = synth.seeded_model(args.seed_surfels, tick=300, seed=args.seed) if hd else None
    head = run_leg(capi, cam, frames, K, Wm, base, events=not args.no_events, seed_model=seed_model, seed_tick=300,
                   mode="device_sync" if args.sync_every_frame else "device")
    log, counts, tim = head["log"], head["counts"], head["tim"]
    kern, roofline = None, None
    if tim is not None:
        kern, _, roofline = kernel_table(log, tim, P, K, Wm, args.compact_period, args.workload)
        roofline["note"] = ("not byte-bound at this size: a frame touches ~60 MB; tools/pass_trace.py (DESIGN.md 4) accounts for the launch. "
                            "hd_leg carries the same figures where the surfel pass IS the frame (20 M surfels)")
    F_total, U_total = float(log["fused_count"].sum()), float(log["unstable_count"].sum())
    elapsed = K / head["fps"]

    legs = {}
    if not hd and not args.no_steady_leg:
        # the fair steady-state figure: the driver's 20-frame window holds one compaction, 100 frames hold six
        r = run_leg(capi, cam, frames, args.steady_steps, args.steady_warmup, base, events=False)
        legs["steady_leg"] = dict(leg_summary(r, args.steady_steps), config=f"the same stream, {args.steady_steps} frames after {args.steady_warmup}",
                                  steps=args.steady_steps, warmup=args.steady_warmup)
    if not hd and not args.no_fuse_leg:
        # frames that actually fuse: 4 mm depth noise and fuse_thresh 0.05 (data.vert:177-208 carries the frame)
        ff = make_frames(cam, Wm + K, args.seed, 4.0, workers)
        r = run_leg(capi, cam, ff, K, Wm, dict(base, fuse_thresh=0.05), events=not args.no_events)
        fl = dict(leg_summary(r, K), config="same KITTI trajectory and scene, depth noise 4 mm, fuse_thresh 0.05 (Config::surfelFuseDistanceThreshFactor)",
                  fused_F_per_sec=float(r["log"]["fused_count"].sum()) / (r["ms"] * 1e-3 * K))
        if r["tim"] is not None:
            fk, _, froof = kernel_table(r["log"], r["tim"], P, K, Wm, args.compact_period, "kitti_fuse")
            fl["kernels"] = fk
            fl["roofline"] = {k: froof[k] for k in ("kernel", "achieved", "frac", "ms_per_launch", "alg_bytes_per_launch")}
        legs["fuse_leg"] = fl
    if not hd and not args.no_reference_path_leg:
        # what the drop-in's default SurfelMapping::processFrame executes (facade/SurfelMapping.h -> sm_process_frame;
        # /root/reference/src/SurfelMapping.cpp:122-156): three host images uploaded, p0a..p0e, the caller waits for the frame
        pre = dict(base, preprocess=1)
        rp = {"config": "preprocess = 1 (p0a..p0e), KITTI 1242x375, the same stream"}
        r = run_leg(capi, cam, frames, K, Wm, pre, events=False, mode="host_sync")
        rp["host_buffers_sync"] = dict(leg_summary(r, K), what="sm_process_frame: 2.8 MB H2D + one host wait per frame (the reference's semantics)")
        if hasattr(capi.SurfelMap, "process_frame_async"):
            r = run_leg(capi, cam, frames, K, Wm, pre, events=False, mode="host_async")
            rp["host_buffers_async"] = dict(leg_summary(r, K), what="sm_process_frame_async: pinned staging, the H2D copy of frame f+1 overlaps frame f, no host wait")
        r = run_leg(capi, cam, frames, K, Wm, pre, events=False, mode="device")
        rp["device_resident_async"] = dict(leg_summary(r, K), what="sm_process_frame_device, frames staged in HBM, no host wait")
        if "host_buffers_async" in rp:
            rp["host_over_device"] = rp["host_buffers_async"]["ms_per_step"] / rp["device_resident_async"]["ms_per_step"]
        legs["reference_path_leg"] = rp
    if not hd and not args.no_hd_leg:
        legs["hd_leg"] = hd_leg(args, capi, workers)

    # ---- CPU baseline: the oracle over the very same frames (rank 0, N = 1 only), a bounded sample
    cpu, cpu_all = None, None
    if not args.no_cpu_baseline:
        import oracle_lib as ol                  # checker / baseline only
        ocfg = dict(preprocess=args.preprocess, conflict_cap=0 if hd else 1, max_sqrt_vertices=10000 if hd else 5000)
        Kc = min(K, 3) if hd else min(K, args.cpu_frames)
        v1, oc, _ = oracle_run(ol, cam, frames, Wm, Kc, ocfg, seed_model=seed_model, seed_tick=300)
        same = None
        if Kc == K:
            same = all(oc[k] == counts[k] for k in ("count", "offset", "unstable_count", "fused_count", "conflict_count"))
        by_pass_1 = oc.pop("_ms_per_frame_by_pass")
        cpu = {"value": v1, "unit": "frames/s", "cores": 1, "kind": "port", "ms_per_frame_by_pass": by_pass_1,
               "sample": f"the first {Kc} of the same {K} frames (after the same {Wm} warm-up frames), oracle/libsmo.so, 1 thread, "
                         f"host {os.cpu_count()} logical CPUs", "final_counts_match_gpu": same}
        try:
            # all cores: the OpenMP build of the same oracle source (bit-identical results: tests/test_oracle_omp.py) on the CPU
            # share of one GPU of this box (256 logical CPUs / 8 GPUs = 32)
            nthr = max(1, min(os.cpu_count() or 1, int(os.environ.get("SM_BENCH_CPU_THREADS", "32"))))
            os.environ["OMP_NUM_THREADS"] = str(nthr)
            va, oac, _ = oracle_run(ol, cam, frames, Wm, Kc, ocfg, libpath=ol.OMP_LIB_PATH, seed_model=seed_model, seed_tick=300)
            by_pass_n = oac.pop("_ms_per_frame_by_pass")
            cpu_all = {"value": va, "unit": "frames/s", "cores": nthr, "kind": "port",
                       "sample": f"the same {Kc} frames, oracle/libsmo_omp.so (OpenMP build of the same source), {nthr} threads "
                                 f"(this box's CPU share of one GPU: {os.cpu_count()} logical CPUs / 8)",
                       "final_counts_match_1_thread": all(oac[k] == oc[k] for k in oc),
                       "ms_per_frame_by_pass": by_pass_n, "speedup_over_1_thread": va / v1}
        except Exception as e:               # the baseline is a report, never a reason to lose the GPU line
            cpu_all = {"error": repr(e)}

    out = {
        "metric": ("frames/sec per GPU, 1920x1080 dense depth, >=20 M live surfels" if hd else METRIC),
        "value": head["fps"], "unit": "frames/s", "n_gpus": 1, "steps": K, "warmup": Wm,
        "ms_per_step": head["ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"BASELINE configs[2]: 1920x1080 dense depth, model pre-seeded with {args.seed_surfels} surfels, conflict cap off, "
                                if hd else "BASELINE configs[1]: KITTI 1242x375 synthetic street sequence, 0.8 m/frame, ")
                               + f"depth noise {args.noise_mm} mm, metricise+conflict+cull+splat+associate+fuse+append per frame"
                               + (" + depth pre-processing p0b..p0e" if args.preprocess else ""),
                   "frames": f"{Wm}..{Wm + K - 1}", "multi_gpu": "single stream",
                   "host_sync": "per frame" if args.sync_every_frame else "none inside the timed region",
                   "compaction": (f"deferred: culled surfels keep their slots, every {ordinal(args.compact_period)} cull compacts "
                                  f"({int((log['n_static'] < log['n_slots']).sum()) if len(log) else 0} of {K} timed frames moved surfels)")
                                 if args.compact_period > 1 else "every frame",
                   "surfels_start": int(log["n_before"][0]) if len(log) else 0, "surfels_end": int(counts["count"])},
        "surfels_fused_per_sec": (F_total + U_total) / elapsed,
        # F (measurements integrated into an existing surfel) and U (new surfels) separately.  With the reference's
        # default fuse threshold 0.0 only bit-equal ray depths associate, so on a moving camera F ~ 0 and the figure above is
        # append throughput; `fuse_leg` times the integration path with F > 0.
        "fused_F": {"total": F_total, "per_sec": F_total / elapsed, "per_frame": F_total / K},
        "new_U": {"total": U_total, "per_sec": U_total / elapsed, "per_frame": U_total / K},
        "conflicts_per_frame": float(log["conflict_count"].mean()) if len(log) else 0.0,
        "roofline": roofline,
        "cpu_baseline": cpu,
        "cpu_baseline_all_cores": cpu_all,
        "kernels": kern,
        "frame_ms_gpu_events": tim["run"] if tim else None,
        "host_enqueue_ms_per_step": head["host_enqueue_ms"],
        "event_overhead_ms": tim.get("event_overhead", 0.0) if tim else None,
        "gen_seconds": t_gen,
    }
    out.update(legs)
    emit(out)


def hd_leg(args, capi, workers):
    """BASELINE configs[2] inside the default line: 1920x1080, model pre-seeded with 20 M surfels, conflict cap off, 40 frames
    after 5; its own roofline entries for k_surfel_pass and k_assoc_prep (the >= 40 % target lives here), PMC traffic from
    profiles/traffic_hd20m_s40_w5.json; the first frames' counters checked against the all-core oracle."""
    cam = synth.HD
    P = cam["width"] * cam["height"]
    K, Wm = args.hd_steps, args.hd_warmup
    t0 = time.time()
    frames = make_frames(cam, Wm + K, args.seed, args.noise_mm, workers)
    seed_model = synth.seeded_model(args.seed_surfels, tick=300, seed=args.seed)
    t_gen = time.time() - t0
    cfg = dict(preprocess=0, conflict_cap=0, max_sqrt_vertices=10000, compact_period=args.compact_period)
    r = run_leg(capi, cam, frames, K, Wm, cfg, events=not args.no_events, seed_model=seed_model, seed_tick=300)
    out = dict(leg_summary(r, K), config=f"BASELINE configs[2]: 1920x1080 dense depth, {args.seed_surfels} seeded surfels (uniformly scattered), "
                                         f"conflict cap off, depth noise {args.noise_mm} mm", steps=K, warmup=Wm, gen_seconds=t_gen)
    if r["tim"] is not None:
        kern, _, roof = kernel_table(r["log"], r["tim"], P, K, Wm, args.compact_period, "hd20m")
        out["kernels"] = kern
        out["roofline"] = roof
        out["roofline_by_kernel"] = {n: {"achieved_GBs": kern[n]["GBs"], "frac": kern[n]["frac_of_hbm_peak"], "ms_per_launch": kern[n]["ms"],
                                         "alg_MB_per_launch": kern[n]["MB"], "hbm_traffic_MB_per_launch": kern[n]["hbm_traffic_MB"]}
                                     for n in ("k_surfel_pass", "k_assoc_prep", "k_compact") if n in kern}
    if not args.no_cpu_baseline:
        try:
            import oracle_lib as ol
            Kc = min(K, args.hd_cpu_frames)
            nthr = max(1, min(os.cpu_count() or 1, int(os.environ.get("SM_BENCH_CPU_THREADS", "32"))))
            os.environ["OMP_NUM_THREADS"] = str(nthr)
            va, oc, seq = oracle_run(ol, cam, frames, Wm, Kc, dict(preprocess=0, conflict_cap=0, max_sqrt_vertices=10000),
                                     libpath=ol.OMP_LIB_PATH, seed_model=seed_model, seed_tick=300, per_frame=True)
            by_pass = oc.pop("_ms_per_frame_by_pass")
            lg = r["log"][:Kc]
            same = all(int(lg["n_after_cull"][k]) == seq[k]["offset"] and int(lg["unstable_count"][k]) == seq[k]["unstable_count"] and
                       int(lg["fused_count"][k]) == seq[k]["fused_count"] and int(lg["conflict_count"][k]) == seq[k]["conflict_count"] and
                       int(lg["visible_count"][k]) == seq[k]["visible_count"] for k in range(Kc))
            out["cpu_baseline_all_cores"] = {"value": va, "unit": "frames/s", "cores": nthr, "kind": "port",
                                             "sample": f"frames {Wm}..{Wm + Kc - 1} of the same stream on the same seeded model, oracle/libsmo_omp.so, {nthr} threads",
                                             "final_counts_match_gpu": bool(same), "ms_per_frame_by_pass": by_pass,
                                             "checked": "offset, new, fused, conflict and index-map counts of every sampled frame against the GPU's frame log"}
        except Exception as e:
            out["cpu_baseline_all_cores"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--noise-mm", type=float, default=15.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=40, help="frames of the CPU-oracle sample (bounded: ~1-2 s of one core each)")
    ap.add_argument("--workers", type=int, default=0, help="frame-generation processes (0 = auto; use 1 under rocprofv3)")
    ap.add_argument("--no-events", action="store_true", help="no HIP events inside frames (no per-kernel timings / roofline)")
    ap.add_argument("--preprocess", type=int, default=0, help="1: run the full depth pre-processing chain p0a..p0e per frame")
    ap.add_argument("--workload", choices=["kitti", "hd20m"], default="kitti",
                    help="kitti = BASELINE configs[1] (default, the metric's config); hd20m = configs[2] as the whole run: 1920x1080 "
                         "dense depth, model pre-seeded with 20 M surfels (HBM-bandwidth stress)")
    ap.add_argument("--seed-surfels", type=int, default=20_000_000)
    ap.add_argument("--force-dist", action="store_true", help="rehearse the multi-GPU code path with one rank (torch.distributed + RCCL up)")
    ap.add_argument("--no-sharded-leg", action="store_true", help="N ranks: skip the ONE-stream-over-N-GPUs leg (configs[3])")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N ranks, all on GPU 0, gloo + host-staged collectives: rehearses the N-process flow on a 1-GPU box (not a measurement)")
    ap.add_argument("--compact-period", type=int, default=24,
                    help="deferred compaction: culled surfels keep their slots, every Nth cull squeezes them out (1: every frame)")
    ap.add_argument("--no-fuse-leg", action="store_true")
    ap.add_argument("--no-steady-leg", action="store_true")
    ap.add_argument("--no-reference-path-leg", action="store_true")
    ap.add_argument("--no-hd-leg", action="store_true")
    ap.add_argument("--only-headline", action="store_true", help="no extra legs (profiling runs)")
    ap.add_argument("--steady-steps", type=int, default=100)
    ap.add_argument("--steady-warmup", type=int, default=10)
    ap.add_argument("--hd-steps", type=int, default=40)
    ap.add_argument("--hd-warmup", type=int, default=5)
    ap.add_argument("--hd-cpu-frames", type=int, default=2)
    ap.add_argument("--sync-every-frame", action="store_true",
                    help="reference semantics: host waits for the counters after every frame")
    ap.add_argument("--launch-dry-run", action="store_true", help="start the ranks, rendezvous over gloo, print who is who; no GPU work")
    ap.add_argument("--launch-timeout", type=float, default=1500.0)
    args = ap.parse_args()
    if args.only_headline:
        args.no_fuse_leg = args.no_steady_leg = args.no_reference_path_leg = args.no_hd_leg = True

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])           # before anything touches a GPU (nothing above imports torch or the core)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: refusing to measure another configuration than the one asked for")
    if args.launch_dry_run:
        return dry_run_rank(rank, world, local_rank)

    # ONE JSON line on stdout: native libraries (RCCL prints a version banner to stdout when a communicator is created)
    # write to file descriptor 1 directly, so fd 1 is pointed at stderr for the whole run and the line goes to the real stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    K, Wm = args.steps, max(args.warmup, 2)     # call 1 only sets the reference frame
    cam = synth.HD if args.workload == "hd20m" else synth.KITTI
    workers = args.workers or max(1, min(16, (os.cpu_count() or 2) // max(world, 1)))
    if world > 1 or args.force_dist:
        return bench_ranks(args, cam, rank, world, local_rank, K, Wm, workers, emit)
    return bench_single(args, cam, K, Wm, workers, emit)


if __name__ == "__main__":
    main()
