#!/usr/bin/env python3
"""bench.py -- frames/sec of the per-frame surfel-fusion hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one SurfelMapping::processFrame of the hot path (metricise -> conflict/cull ->
index-map splat -> associate/fuse -> append) over one 1242x375 KITTI-shaped synthetic
RGB-D+semantic frame (BASELINE.json configs[1]); frames are resident in HBM before the timed
region starts and are enqueued back-to-back (no host read-back inside a frame or between
frames -- the counters of every frame are audited afterwards from the device-side frame log).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against the 8 TB/s HBM
peak with algorithmic bytes from the per-frame counters (DESIGN.md "Measurement");
`cpu_baseline` is the CPU oracle (oracle/, a scalar restatement of the reference's passes)
timed on this host over the same frames, 1 thread.
"""
from __future__ import annotations

import argparse
import gc
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from surfelmapping_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def rig_trajectory(n, rank, world, step=0.8):
    """Camera `rank` of a `world`-camera rig: same forward motion, yaw offset rank*360/world deg
    (BASELINE configs[4] / SURVEY 8d config 5); world == 1 is the plain KITTI trajectory."""
    import math
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
    # exact count is not logged, this is the lower bound), the time (4) of the slots whose tile reaches the index map,
    # one key atomic per drawn surfel
    # (lane-compacting form, the default: the time plane is read only for the slots that reach the exact tests -- not
    #  logged; lower bound: the drawn ones.  SM_PASS_COMPACT=0, word by word: every slot of a tile that reaches the index map)
    t_read = V if os.environ.get("SM_PASS_COMPACT", "1") != "0" else np.maximum(Sl - Ss, 0.0)
    one_pass = 16.0 * np.maximum(Sl - np.minimum(Cs, Ss), 0.0) + 4.0 * t_read + 8.0 * V
    return {
        "k_prep": np.full_like(N, 6.0 * P + 24.0 * P),          # u8x3+u16+u8 in, f32+u32+u64+(f32,u32) out
        "k_conflict": 16.0 * np.maximum(Sl - Cs, 0.0),            # tiles skipped by their bounds are not read
        "k_compact": compact,
        "k_surfel_pass": one_pass,
        "k_pass_fixup": np.full_like(N, 4.0 * P),                 # the candidate count reads the depth plane once
        "k_associate": 16.0 * P + 84.0 * F,                       # depth+rgbs+key per pixel, gather 44 + scatter 40 per fuse
        "k_append": 8.0 * (P / 64.0) + 44.0 * U,
        "k_associate_direct": 16.0 * P + 84.0 * F + 44.0 * U,     # ... and the new surfels written in place
    }


def kernel_table(log, tim, P, K, args, workload=None):
    """Per-kernel launches, average duration (HIP events), algorithmic bytes and GB/s of the timed frames, and the
    roofline entry of the kernel the run spends most time in.  The frame forms: a frame whose cull only
    marks the dead runs k_surfel_pass + k_pass_fixup (+ k_associate_direct when it appends directly), the others
    k_conflict + k_scan_cull/k_cull_finalize + k_compact (or k_cull_lazy) and k_associate + k_append_scan."""
    logd = {k: log[k] for k in log.dtype.names}
    logd["P"] = P
    kb = kernel_bytes(logd)
    n_op, n_dir, n_comp = int(tim.get("frames_one_pass", 0)), int(tim.get("frames_direct", 0)), int(tim.get("frames_compact", 0))
    moved = log["n_static"] < log["n_slots"] if len(log) else np.zeros(0, bool)      # frames that compacted
    # (one-pass <=> not compacted on the default path; direct <=> one-pass and k_prep evaluated the tile flags)
    sel_op = ~moved if n_op else np.zeros(len(log), bool)
    sel_dir = sel_op if n_dir == n_op else np.zeros(len(log), bool)
    # asynchronous plain streams: a frame's association is held back and runs in the NEXT frame's preparation launch (k_assoc_prep)
    n_mrg, n_alone = int(tim.get("frames_merged", 0)), int(tim.get("frames_assoc_alone", n_dir))
    sel_mrg = np.zeros(len(log), bool)
    if n_mrg and len(log):
        sel_mrg[1:] = sel_dir[:-1]                        # frame k's launch carries frame k-1's association
        sel_mrg[0] = n_mrg > int(sel_mrg.sum())           # (the last warm-up frame's, if the counts say so)
        kb["k_assoc_prep"] = kb["k_prep"] + np.concatenate([kb["k_associate_direct"][:1], kb["k_associate_direct"][:-1]])
    rows = [("k_prep", tim.get("k_prep_own", tim["k_prep"]) if n_mrg else tim["k_prep"], ~sel_mrg, K - n_mrg),
            ("k_assoc_prep", tim.get("k_assoc_prep", 0.0), sel_mrg, n_mrg),
            ("k_surfel_pass", tim.get("k_surfel_pass", 0.0), sel_op, n_op),
            ("k_pass_fixup", tim.get("k_pass_fixup", 0.0), sel_op, n_op),
            ("k_associate_direct", tim.get("k_associate_direct", 0.0), sel_dir, n_alone),
            ("k_conflict", tim.get("k_conflict_own", 0.0), ~sel_op, K - n_op),
            ("k_compact", tim.get("k_compact_own", 0.0), moved, n_comp),
            ("k_associate", tim.get("k_associate_own", 0.0), ~sel_dir, K - n_dir),
            ("k_append", tim.get("k_append_own", 0.0), ~sel_dir, K - n_dir)]
    if K - n_op - n_comp > 0:      # lazy culls outside the one-pass form (SM_ONE_PASS=0, or the depth filter chain on the second stream)
        rows.append(("k_cull_lazy", tim.get("k_cull_lazy", 0.0), ~moved & ~sel_op, K - n_op - n_comp))
    kern, launches = {}, {}
    for name, ms, sel, n in rows:
        if n <= 0:
            continue
        b = kb["k_compact" if name == "k_cull_lazy" else name]
        mb = float(b[sel].mean()) / 1e6 if len(log) and sel.any() else 0.0
        kern[name] = {"ms": ms, "MB": mb, "GBs": (mb / 1e3) / (ms * 1e-3) if ms > 0 else None, "launches": n}
        launches[name] = n
    if K - n_op > 0:
        kern["k_scan_cull+k_cull_finalize"] = {"ms": tim["k_scan_cull"] * K / max(K - n_op, 1), "MB": None, "GBs": None, "launches": K - n_op}
    # dominant kernel = the one the run spends most time in (average duration x launches)
    dom = max(launches.keys(), key=lambda n: kern[n]["ms"] * launches[n])
    achieved = kern[dom]["GBs"] or 0.0
    traffic, valu = None, None
    wl = workload or args.workload
    for tpath in (os.path.join(ROOT, "profiles", f"traffic_{wl}_s{K}_w{max(args.warmup, 2)}.json"), os.path.join(ROOT, "profiles", f"traffic_{wl}.json")):
        if not os.path.exists(tpath):
            continue
        # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very
        # command (profiles/README.md), gfx950-corrected per kernel by tools/prof_summary.py
        tj = json.load(open(tpath))
        if tj.get("steps") == K and tj.get("warmup") == max(args.warmup, 2) and tj.get("compact_period", 16) == args.compact_period:
            traffic = tj.get("kernels", {}).get(dom, {}).get("hbm_bytes_per_launch")
            valu = tj.get("kernels", {}).get(dom, {}).get("valu_issue_util")
            break
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "ms_per_launch": kern[dom]["ms"], "alg_bytes_per_launch": kern[dom]["MB"] * 1e6,
                "launches": launches[dom],
                "valu_issue_util": valu,
                "note": "not byte-bound: a frame touches ~60 MB.  tools/pass_trace.py (per-workgroup time stamps of one launch, DESIGN.md 4): "
                        "the 2 048 workgroups of k_surfel_pass take 5-8 us to enter the chip, every visited tile's 16 KB arrives in one "
                        "~4 us burst at the start, and the ~240 tiles that hold most of the surfels in view then run ~350 IEEE-exact "
                        "VALU instructions per surfel for ~6 us; k_assoc_prep is one such burst (28 MB) plus two dependent gathers"}
    return kern, launches, roofline


def bench_sharded(args, dist, torch, capi, cam, frames, rank, world, local_rank, K, Wm, t_gen, emit):
    """ONE camera stream split over the GPUs (BASELINE configs[3]); every rank holds the replicated frame sequence (same
    seed, no rank offset).  Default: the in-stream form (sm_shard_frame_device: slot-addressed shards, RCCL called from the
    HIP core on the context's stream, frames resident in HBM); --shard-form staged is round 1's per-stage form driven from
    Python (surfelmapping_amd/sharded.py ShardedMapper), kept for comparison."""
    from surfelmapping_amd import sharded
    P = cam["width"] * cam["height"]
    stream = args.shard_form == "stream"
    kw = dict(preprocess=args.preprocess, device=local_rank, conflict_cap=1)
    if args.compact_period is not None:
        kw["compact_period"] = args.compact_period
    sm = capi.SurfelMap(capi.make_config(**cam, **kw))
    if stream:
        ids = [capi.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        mp = sharded.StreamShard(sm, rank, world, *((None,) if (args.no_rccl and world == 1) else ("rccl", ids[0])))
        dptr = []
        for rgb, depth, sem, pose in frames:
            dr, dd, ds = sm.device_alloc(P * 3), sm.device_alloc(P * 2), sm.device_alloc(P)
            sm.device_upload(dr, rgb); sm.device_upload(dd, depth); sm.device_upload(ds, sem)
            dptr.append((dr, dd, ds, pose))
        step = lambda k: sm.shard_frame_device(*dptr[k])
    else:
        mp = sharded.ShardedMapper(sharded.HipShardBackend(sm, rank, world), sharded.TorchComm(device_index=local_rank), P,
                                   collect_stats=False)
        step = lambda k: mp.process_frame(*frames[k])
    # no collector pause inside the timed region: with torch imported a full collection takes ~45 ms (measured: one frame call
    # of 110 stalled that long).  Collect BEFORE the warm-up: a 45 ms pause between warm-up and t0 lets the GPU clock down.
    gc.collect(); gc.disable()
    for k in range(Wm):
        step(k)
    sm.sync()
    c0 = sm.counts() if stream else mp.counts()
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    enq = []
    for k in range(Wm, Wm + K):
        te = time.perf_counter()
        step(k)
        enq.append(time.perf_counter() - te)
    t_enq = time.perf_counter() - t0
    sm.sync(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0      # this rank's K frames are complete; the MAX over the ranks below is when the last rank was
    dist.barrier(); torch.cuda.synchronize()  # (the closing barrier itself -- ~0.4 ms with torch's NCCL -- is not part of the K frames)
    gc.enable()
    t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cuda", local_rank))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0])
    c1 = sm.counts() if stream else mp.counts()
    log = sm.read_frame_log(K) if stream else None
    fused = int(log["fused_count"].sum() + log["unstable_count"].sum()) if log is not None and len(log) else 0
    plain = None
    if rank == 0 and stream and not args.no_plain_leg:
        # the same frames on the plain single-GPU path of this GPU: what sharding has to beat
        fps1, ms1, _, cp, _ = run_simple_leg(capi, cam, frames, K, Wm, dict(preprocess=args.preprocess, conflict_cap=1, **(
            {"compact_period": args.compact_period} if args.compact_period is not None else {})), argparse.Namespace(no_events=True))
        plain = {"frames_per_sec": fps1, "ms_per_step": ms1, "surfels_end": int(cp["count"]),
                 "sharded_over_plain": (elapsed / K * 1e3) / ms1}
    if stream and not (args.no_rccl and world == 1):
        sm.shard_rccl_finalize()
    dist.destroy_process_group()
    if rank != 0:
        return
    form = ("in-stream form: slot-addressed shards, per frame all-reduce(min) of the 3.7 MB key map and all-reduce(sum) of the fused-pixel "
            "mask + 3 counters, RCCL called from the HIP core on the frame's stream, frames resident in HBM, deferred compaction "
            "between frames (all-reduce(sum) of the alive bits)") if stream else (
            "staged form driven from Python: all-reduce(sum) segment counts, all-reduce(min) key map, all-reduce(sum) fused mask; "
            "host-buffer frames (PCIe-inclusive)")
    emit(({
        "metric": "frames/sec, 1242x375 KITTI-shaped RGB-D+semantic, full associate+fuse+merge",
        "value": K / elapsed, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3]: ONE KITTI 1242x375 stream sharded over {world} GPUs, {form}; result bit-identical to 1 GPU",
                   "frames": f"{Wm}..{Wm + K - 1}", "surfels_start": int(c0["count"]), "surfels_end": int(c1["count"])},
        "surfels_fused_per_sec": fused / elapsed, "plain_single_gpu": plain,
        "host_enqueue_ms_per_step": t_enq / K * 1e3, "host_enqueue_max_ms": max(enq) * 1e3, "host_enqueue_argmax": int(np.argmax(enq)), "host_enqueue_median_ms": float(np.median(enq)) * 1e3, "roofline": None, "cpu_baseline": None, "gen_seconds": t_gen}))


def run_simple_leg(capi, cam, frames, K, Wm, cfg_kw, args):
    """One single-GPU leg: stage the frames in HBM, time K frames after Wm on an un-instrumented context, replay them on
    a context with HIP events for the per-kernel durations.  Returns (frames/s, ms/step, frame log, counts, timings)."""
    P = cam["width"] * cam["height"]

    def stage(sm):
        out = []
        for rgb, depth, sem, pose in frames:
            dr, dd, ds = sm.device_alloc(P * 3), sm.device_alloc(P * 2), sm.device_alloc(P)
            sm.device_upload(dr, rgb); sm.device_upload(dd, depth); sm.device_upload(ds, sem)
            out.append((dr, dd, ds, pose))
        return out

    sm = capi.SurfelMap(capi.make_config(**cam, **cfg_kw, enable_timing=0))
    dptr = stage(sm)
    gc.collect(); gc.disable()
    for k in range(Wm):
        sm.process_frame_device(*dptr[k])
    sm.sync()
    t0 = time.perf_counter()
    for k in range(Wm, Wm + K):
        sm.process_frame_device(*dptr[k])
    sm.sync()
    el = time.perf_counter() - t0
    gc.enable()
    log, counts = sm.read_frame_log(K), sm.counts()
    sm.close()
    tim = None
    if not args.no_events:
        sm = capi.SurfelMap(capi.make_config(**cam, **cfg_kw, enable_timing=1))
        dptr = stage(sm)
        for k in range(Wm):
            sm.process_frame_device(*dptr[k])
        sm.sync(); sm.timings()
        for k in range(Wm, Wm + K):
            sm.process_frame_device(*dptr[k])
        sm.sync()
        tim = sm.timings()
        sm.close()
    return K / el, el / K * 1e3, log, counts, tim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--noise-mm", type=float, default=15.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workers", type=int, default=0, help="frame-generation processes (0 = auto; use 1 under rocprofv3)")
    ap.add_argument("--no-events", action="store_true", help="no HIP events inside frames (no per-kernel timings / roofline)")
    ap.add_argument("--preprocess", type=int, default=0, help="1: run the full depth pre-processing chain p0a..p0e per frame")
    ap.add_argument("--workload", choices=["kitti", "hd20m"], default="kitti",
                    help="kitti = BASELINE configs[1] (default, the metric's config); hd20m = configs[2]: 1920x1080 "
                         "dense depth, model pre-seeded with 20 M surfels (HBM-bandwidth stress)")
    ap.add_argument("--seed-surfels", type=int, default=20_000_000)
    ap.add_argument("--mode", choices=["rig", "sharded"], default="rig",
                    help="N>1: 'rig' = one camera stream per GPU + all-gather into one GlobalModel (weak scaling); "
                         "'sharded' = ONE stream split over the GPUs, bit-identical to 1 GPU (strong scaling)")
    ap.add_argument("--force-dist", action="store_true", help="rehearse the multi-GPU code path with WORLD_SIZE=1")
    ap.add_argument("--shard-form", choices=["stream", "staged"], default="stream",
                    help="--mode sharded: in-stream form (RCCL from the HIP core; default) or round 1's per-stage form driven from Python")
    ap.add_argument("--no-rccl", action="store_true", help="--mode sharded with one rank: no communicator (the core's identity path) instead of RCCL")
    ap.add_argument("--no-plain-leg", action="store_true", help="--mode sharded: skip the plain single-GPU run of the same frames on rank 0")
    ap.add_argument("--compact-period", type=int, default=16,
                    help="deferred compaction: culled surfels keep their slots, every Nth cull squeezes them out (1: every frame)")
    ap.add_argument("--no-fuse-leg", action="store_true",
                    help="skip the second, labelled leg (same trajectory, depth noise 4 mm, fuse_thresh 0.05: frames that actually fuse)")
    ap.add_argument("--sync-every-frame", action="store_true",
                    help="reference semantics: host waits for the counters after every frame")
    args = ap.parse_args()

    # ONE JSON line on stdout: native libraries (RCCL prints a version banner to stdout when a communicator is created)
    # write to file descriptor 1 directly, so fd 1 is pointed at stderr for the whole run and the line goes to the real stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    K, Wm = args.steps, max(args.warmup, 2)     # call 1 only sets the reference frame
    hd = args.workload == "hd20m"
    cam = synth.HD if hd else synth.KITTI
    P = cam["width"] * cam["height"]
    n_frames = Wm + K

    # ---- synthetic frames (before anything touches the GPU; forked workers never do)
    t0 = time.time()
    workers = args.workers or max(1, min(8, (os.cpu_count() or 2) // max(world, 1)))
    shard = args.mode == "sharded" and (world > 1 or args.force_dist)
    frames = make_frames(cam, n_frames, args.seed, args.noise_mm, workers, 0 if shard else rank, 1 if shard else world)
    t_gen = time.time() - t0

    dist = None
    if world > 1 or args.force_dist:
        # torch first: its HIP runtime must be the one the core binds to (one runtime per process)
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:                     # --force-dist outside torchrun: a one-rank rendezvous of its own
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from surfelmapping_amd import capi
    if dist and args.mode == "sharded":
        return bench_sharded(args, dist, torch, capi, cam, frames, rank, world, local_rank, K, Wm, t_gen, emit)
    # hd20m: the conflict cap is off for the stress benchmark (SURVEY.md A13 says to state which)
    mk = lambda timing: capi.SurfelMap(capi.make_config(**cam, preprocess=args.preprocess, device=local_rank,
                                                        enable_timing=timing, conflict_cap=0 if hd else 1,
                                                        max_sqrt_vertices=10000 if hd else 5000,
                                                        compact_period=args.compact_period))
    # this process creates a second (instrumented) context later, but the two never run at the same time: keep the
    # default round-robin form of the compaction kernel for both (the library would switch to ticket order otherwise)
    os.environ.setdefault("SM_COMPACT_TICKETS", "0")
    sm = mk(0)                                    # raises without a GPU: no CPU fallback
    # A second context replays the same frames with HIP events between the kernels (the events cost ~25 us per frame,
    # so they stay out of the run that produces `value`).  It is created AFTER the timed run: the timed context is then
    # the only one on its GPU, as in deployment (contexts that share a GPU chain their compaction kernels).
    sm_ev = None

    # ---- stage every frame in HBM
    dptr = []
    for rgb, depth, sem, pose in frames:
        dr, dd, ds = sm.device_alloc(P * 3), sm.device_alloc(P * 2), sm.device_alloc(P)
        sm.device_upload(dr, rgb); sm.device_upload(dd, depth); sm.device_upload(ds, sem)
        dptr.append((dr, dd, ds, pose))

    seed_model = None
    if hd:
        seed_model = synth.seeded_model(args.seed_surfels, tick=300, seed=args.seed)
        sm.upload_model(seed_model)
        sm.set_tick(300)

    def run(ctx, lo, hi):
        for k in range(lo, hi):
            ctx.process_frame_device(*dptr[k])
            if args.sync_every_frame:
                ctx.sync()

    sm_global = None
    if dist:
        from surfelmapping_amd import dist as smd
        # warm the collective path (RCCL communicator setup is not part of a frame)
        g0, c0 = smd.gather_model_device(sm, local_rank)
        del g0

    def barrier():
        if dist:
            dist.barrier()
            torch.cuda.synchronize()

    gc.collect(); gc.disable()                    # no collector pause inside the timed region (~45 ms with torch imported), and none
    run(sm, 0, Wm)                                # between the warm-up and t0 either (the GPU would clock down: +130 us on the first frames)
    sm.sync()
    barrier()
    t0 = time.perf_counter()
    run(sm, Wm, Wm + K)
    t_enq = time.perf_counter() - t0              # host time to enqueue the K frames (no waiting inside)
    sm.sync()
    if dist:
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0            # this rank's K frames are complete (MAX over the ranks is taken below)
    barrier()                                     # the closing barrier + synchronize; its own latency (~0.4 ms) is not K frames' work
    gc.enable()
    # Consolidation into a single GlobalModel (BASELINE configs[4]): an end-of-run exchange, not part of a frame --
    # the per-frame hot path of a camera touches only its own slice -- so it is timed separately.
    global_count, gather_ms = None, None
    if dist:
        from surfelmapping_amd import sharded as smsh
        sm_global = capi.SurfelMap(capi.make_config(**cam, preprocess=0, device=local_rank, max_sqrt_vertices=10000 if (hd or world > 2) else 5000))
        rig = smd.RigMapper(sm, smsh.TorchComm(device_index=local_rank), P)
        rig.last = (frames[Wm + K - 1][1], frames[Wm + K - 1][2], frames[Wm + K - 1][3])     # the camera's latest view
        # the consolidation runs inside the HIP core (sm_rig_consolidate), its exchanges on RCCL bound by the core itself;
        # torch.distributed only hands the communicator id round
        ids = [capi.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        rig.enable_native("rccl", ids[0])
        barrier()
        g0 = time.perf_counter()
        # every slice cleaned against every camera's latest view (cleanPoints per view), then the slices gathered in rank order
        global_count, view_conflicts = rig.consolidate_native(sm_global)
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
        sm.shard_rccl_finalize()
    log = sm.read_frame_log(K)
    counts = sm.counts()

    # ---- the same K frames on the instrumented context: per-kernel durations for the roofline
    tim = {k: 0.0 for k in ("k_prep", "k_conflict", "k_scan_cull", "k_compact", "k_associate", "k_scan_new", "k_append", "run")}
    if not args.no_events:
        sm_ev = mk(1)
        if hd:
            sm_ev.upload_model(seed_model)
            sm_ev.set_tick(300)
        run(sm_ev, 0, Wm)
        sm_ev.sync()
        sm_ev.timings()                           # drop warm-up samples
        run(sm_ev, Wm, Wm + K)
        sm_ev.sync()
        tim = sm_ev.timings()
        assert sm_ev.counts()["count"] == counts["count"], "instrumented pass diverged from the timed pass"

    if dist:
        dev = torch.device("cuda", local_rank)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        fu = torch.tensor([float(log["fused_count"].sum()), float(log["unstable_count"].sum())], dtype=torch.float64, device=dev)
        dist.all_reduce(fu, op=dist.ReduceOp.SUM)
        F_total, U_total = float(fu[0]), float(fu[1])
        dist.destroy_process_group()
    else:
        F_total, U_total = float(log["fused_count"].sum()), float(log["unstable_count"].sum())
    fused_total = F_total + U_total

    if rank != 0:
        return

    # ---- roofline of the dominant kernel (live HIP-event durations, algorithmic bytes)
    kern, launches, roofline = kernel_table(log, tim, P, K, args)

    # ---- second, labelled leg: the same trajectory with 4 mm depth noise and fuse_thresh = 0.05, so that the in-place
    # integration (depth/colour/normal/radius update, data.vert:177-208) is timed with F > 0; `value` stays the default config
    fuse_leg = None
    if not hd and dist is None and not args.no_fuse_leg:
        ff = make_frames(cam, n_frames, args.seed, 4.0, workers)
        fv, fms, flog, fcounts, ftim = run_simple_leg(capi, cam, ff, K, Wm, dict(preprocess=args.preprocess, device=local_rank, conflict_cap=1,
                                                                                 fuse_thresh=0.05, compact_period=args.compact_period), args)
        fuse_leg = {"config": "same KITTI trajectory and scene, depth noise 4 mm, fuse_thresh 0.05 (Config::surfelFuseDistanceThreshFactor)",
                    "value": fv, "unit": "frames/s", "ms_per_step": fms,
                    "fused_F_per_frame": float(flog["fused_count"].mean()), "new_U_per_frame": float(flog["unstable_count"].mean()),
                    "fused_F_per_sec": float(flog["fused_count"].sum()) / (fms * 1e-3 * K),
                    "conflicts_per_frame": float(flog["conflict_count"].mean()), "surfels_end": int(fcounts["count"])}
        if ftim is not None:
            fk, _, froof = kernel_table(flog, ftim, P, K, args, workload="kitti_fuse")
            fuse_leg["kernels"] = fk
            fuse_leg["roofline"] = {k: froof[k] for k in ("kernel", "achieved", "frac", "ms_per_launch", "alg_bytes_per_launch")}

    # ---- CPU baseline: the oracle over the very same frames (rank 0, N=1 only)
    cpu, cpu_all = None, None
    if not args.no_cpu_baseline and dist is None:
        import oracle_lib as ol                  # checker / baseline only
        o = ol.Oracle(ol.make_config(**cam, preprocess=args.preprocess, conflict_cap=0 if hd else 1,
                                     max_sqrt_vertices=10000 if hd else 5000))
        Kc = K
        if hd:                                   # bounded sample: a 20 M-surfel frame takes seconds on one core
            o.upload_model(seed_model); o.set_tick(300)
            Kc = min(K, 3)
        for k in range(Wm):
            o.process_frame(*frames[k])
        c0 = time.perf_counter()
        for k in range(Wm, Wm + Kc):
            o.process_frame(*frames[k])
        c_el = time.perf_counter() - c0
        oc = o.counts()
        same = (Kc == K) and all(oc[k] == counts[k] for k in ("count", "offset", "unstable_count", "fused_count", "conflict_count"))
        # all cores: the OpenMP build of the same oracle source (bit-identical results: tests/test_oracle_omp.py), on the
        # CPU share of one GPU of this box (16 hardware threads of 256), same frames
        cpu_all = None
        try:
            nthr = max(1, min(os.cpu_count() or 1, int(os.environ.get("SM_BENCH_CPU_THREADS", "16"))))
            os.environ["OMP_NUM_THREADS"] = str(nthr)
            oa = ol.Oracle(ol.make_config(**cam, preprocess=args.preprocess, conflict_cap=0 if hd else 1,
                                          max_sqrt_vertices=10000 if hd else 5000), libpath=ol.OMP_LIB_PATH)
            if hd:
                oa.upload_model(seed_model); oa.set_tick(300)
            for k in range(Wm):
                oa.process_frame(*frames[k])
            a0 = time.perf_counter()
            for k in range(Wm, Wm + Kc):
                oa.process_frame(*frames[k])
            a_el = time.perf_counter() - a0
            oac = oa.counts()
            cpu_all = {"value": Kc / a_el, "unit": "frames/s", "cores": nthr, "kind": "port",
                       "sample": f"the same {Kc} frames, oracle/libsmo_omp.so (OpenMP build of the same source), {nthr} threads",
                       "final_counts_match_1_thread": all(oac[k] == oc[k] for k in oc)}
            oa.close()
        except Exception as e:               # the baseline is a report, never a reason to lose the GPU line
            cpu_all = {"error": repr(e)}
        cpu = {"value": Kc / c_el, "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": f"the first {Kc} of the same {K} frames (after the same {Wm} warm-up frames), oracle/libsmo.so, 1 thread, "
                         f"host {os.cpu_count()} logical CPUs", "final_counts_match_gpu": bool(same) if Kc == K else None}

    out = {
        "metric": ("frames/sec per GPU, 1920x1080 dense depth, >=20 M live surfels" if hd else
                   "frames/sec per GPU, 1242x375 KITTI-shaped RGB-D+semantic, full associate+fuse+merge"),
        "value": world * K / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": K,
        "warmup": Wm,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": (f"BASELINE configs[2]: 1920x1080 dense depth, model pre-seeded with {args.seed_surfels} surfels, conflict cap off, "
                                if hd else "BASELINE configs[1]: KITTI 1242x375 synthetic street sequence, 0.8 m/frame, ")
                               + f"depth noise {args.noise_mm} mm, metricise+conflict+cull+splat+associate+fuse+append per frame"
                               + (" + depth pre-processing p0b..p0e" if args.preprocess else ""),
                   "frames": f"{Wm}..{Wm + K - 1}", "multi_gpu": (f"{world}-camera rig, one stream per GPU; consolidation after the timed frames: every slice cleaned "
                                 f"against all {world} latest views (cleanPoints per view, {sum(view_conflicts)} conflicts), slices gathered in rank order into a single "
                                 f"GlobalModel of {global_count} surfels, all inside the HIP core over RCCL (sm_rig_consolidate: {gather_ms:.1f} ms, not in `value`)") if dist else "single stream",
                   "host_sync": "per frame" if args.sync_every_frame else "none inside the timed region",
                   "frame_form": ("four launches per frame" if (args.preprocess or args.sync_every_frame or os.environ.get("SM_DEFER_ASSOC", "1") == "0") else
                                  "three launches per frame (k_assoc_prep = the previous frame's association + this frame's image "
                                  "preparation, then k_surfel_pass, k_pass_fixup; DESIGN.md 4)"),
                   "compaction": (f"deferred: culled surfels keep their slots, every {args.compact_period}th cull compacts "
                                  f"({int((log['n_static'] < log['n_slots']).sum()) if len(log) else 0} of {K} timed frames moved surfels)")
                                 if args.compact_period > 1 else "every frame",
                   "surfels_start": int(log["n_before"][0]) if len(log) else 0, "surfels_end": int(counts["count"])},
        "surfels_fused_per_sec": fused_total / elapsed,
        # F (measurements integrated into an existing surfel) and U (new surfels) separately.  With the reference's
        # default fuse threshold 0.0 only bit-equal ray depths associate, so on a moving camera F ~ 0 and the figure above is
        # append throughput; `fuse_leg` below times the integration path with F > 0.
        "fused_F": {"total": F_total, "per_sec": F_total / elapsed, "per_frame": F_total / (K * world)},
        "new_U": {"total": U_total, "per_sec": U_total / elapsed, "per_frame": U_total / (K * world)},
        "conflicts_per_frame": float(log["conflict_count"].mean()) if len(log) else 0.0,
        "fuse_leg": fuse_leg,
        "roofline": roofline,
        "cpu_baseline": cpu,
        "cpu_baseline_all_cores": cpu_all if cpu is not None else None,
        "kernels": kern,
        "frame_ms_gpu_events": tim["run"],
        "host_enqueue_ms_per_step": t_enq / K * 1e3,
        "event_overhead_ms": tim.get("event_overhead", 0.0),
        "gen_seconds": t_gen,
    }
    emit(out)


if __name__ == "__main__":
    main()
