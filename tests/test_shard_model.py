"""One camera stream sharded over G ranks must reproduce the single-process result bit for bit.  CPU: G threads and 2 gloo
processes with oracle-backed ranks (tests/shard_model.py: segment ownership, global ids, the key-map MIN-reduction, disjoint
fused masks, the owner's append) against ONE oracle -- the N > 1 algorithm at world size 2 without a GPU.  The product's form
(slot-addressed, in-stream, RCCL inside the HIP core) is tests/test_shard_stream.py / test_dist_gpu.py with -m gpu."""
import math
import os
import socket
import sys
import threading

import numpy as np
import pytest

import oracle_lib as ol
from backends import assert_models_equal
from shard_model import HostComm, OracleShardBackend, SegmentShardModel
from surfelmapping_amd import sharded, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAM = dict(width=192, height=80, fx=110.0, fy=110.0, cx=95.5, cy=39.5)
OVER = dict(preprocess=0, stereo_border=10.0, fuse_thresh=0.04, conflict_cap=1)
COUNT_KEYS = ("count", "offset", "data_count", "conflict_count", "unstable_count", "fused_count", "visible_count", "tick")


def sequence(n=7, seed=41):
    poses = [synth.pose_matrix(0.02 * k, 0, 0.25 * k, 0.4 * math.sin(k)) for k in range(n)]
    return synth.make_sequence(CAM, poses, seed=seed, noise_mm=3.0)


def reference(seq, **over):
    o = ol.Oracle(ol.make_config(**CAM, **{**OVER, **over}))
    hist = []
    for fr in seq:
        o.process_frame(*fr)
        hist.append({k: o.counts()[k] for k in COUNT_KEYS})
    return o.download_model(), hist


def run_threads(world, seq, make_backend):
    group = sharded.ThreadGroup(world)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            comm = HostComm(sharded.ThreadComm(group, r))
            mp = SegmentShardModel(make_backend(r, world), comm, CAM["width"] * CAM["height"])
            hist = []
            for fr in seq:
                c = mp.process_frame(*fr)
                hist.append({k: c[k] for k in COUNT_KEYS})
            out[r] = (mp.gather_global_model(), hist, list(mp.cnt))
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
            group.barrier.abort()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    return out


@pytest.mark.parametrize("world", [1, 2, 3])
def test_sharded_oracle_ranks_equal_single_process(world):
    seq = sequence()
    ref_model, ref_hist = reference(seq)
    res = run_threads(world, seq, lambda r, w: OracleShardBackend(ol.make_config(**CAM, **OVER), r, w))
    assert ref_hist[-1]["fused_count"] > 50 and ref_hist[-1]["conflict_count"] > 0 and ref_model.shape[0] > 3000
    for model, hist, cnt in res:
        assert hist == ref_hist
        assert_models_equal(model, ref_model, f"world={world}")
        assert sum(cnt) == ref_model.shape[0] and len(cnt) == len(seq) - 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _gloo_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        seq = sequence(5)
        comm = HostComm(sharded.TorchComm(device_index=None))
        mp = SegmentShardModel(OracleShardBackend(ol.make_config(**CAM, **OVER), rank, world), comm,
                                   CAM["width"] * CAM["height"])
        for fr in seq:
            mp.process_frame(*fr)
        q.put((rank, mp.gather_global_model().view(np.uint32).tobytes(), mp.counts()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_two_gloo_processes():
    import torch.multiprocessing as tmp
    ctx = tmp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref_model, ref_hist = reference(sequence(5))
    for rank, blob, counts in res:
        got = np.frombuffer(blob, np.uint32).reshape(-1, 12)
        assert np.array_equal(got, ref_model.view(np.uint32))
        assert {k: counts[k] for k in COUNT_KEYS} == ref_hist[-1]
