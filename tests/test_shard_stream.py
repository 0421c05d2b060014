"""In-stream form of the sharded mode (BASELINE configs[3]): slot-addressed shards, the frame entirely inside
the HIP core, collectives on the context's stream.  G ranks = G HIP contexts on the one GPU driven by G threads with a
host-staged collective (the multi-rank logic), and RCCL itself with world = 1 (the production binding); every variant must
equal the single-process oracle bit for bit -- counters after every frame, the union at the end."""
import threading

import numpy as np
import pytest

import oracle_lib as ol
from surfelmapping_amd import capi, sharded, synth

CAM = dict(width=192, height=80, fx=110.0, fy=110.0, cx=95.5, cy=39.5)


def _seq(n, noise_mm=3.0, seed=11):
    return synth.make_sequence(CAM, synth.kitti_trajectory(n), seed=seed, noise_mm=noise_mm)


def _oracle(seq, over):
    o = ol.Oracle(ol.make_config(**CAM, **over))
    cs = []
    for fr in seq:
        o.process_frame(*fr)
        cs.append(o.counts())
    return o.download_model(), cs


KEYS = ("count", "offset", "conflict_count", "unstable_count", "fused_count", "data_count", "visible_count")


def _run_threads(G, seq, over, period, cam=None):
    grp = sharded.ThreadGroup(G)
    out, errs = [None] * G, []

    def work(r):
        try:
            sm = capi.SurfelMap(capi.make_config(**(cam or CAM), **over, compact_period=period))
            coll = sharded.ThreadCollective(grp, r, sm) if G > 1 else None
            mp = sharded.StreamShard(sm, r, G, coll)
            cs = [mp.process_frame(*fr) for fr in seq]
            out[r] = (mp.export_dense(), cs, sm.counts())
            sm.close()
        except Exception as e:          # keep the other ranks from waiting forever
            errs.append((r, repr(e)))
            grp.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errs, errs
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("G,period", [(1, 8), (2, 3), (4, 3), (3, 1000), (2, 1)])
def test_stream_shards_equal_oracle(G, period):
    over = dict(preprocess=0, stereo_border=10.0, max_sqrt_vertices=400, fuse_thresh=0.05)
    seq = _seq(12)
    ref, ref_counts = _oracle(seq, over)
    out = _run_threads(G, seq, over, period)
    for r in range(G):
        for f, (a, b) in enumerate(zip(out[r][1], ref_counts)):
            assert all(a[k] == b[k] for k in KEYS), (r, f, {k: (a[k], b[k]) for k in KEYS})
    union = sharded.StreamShard.union([o[0] for o in out])
    assert union.shape == ref.shape and ref.shape[0] > 1000
    assert np.array_equal(union.view(np.uint32), ref.view(np.uint32))
    assert sum(c["fused_count"] for c in ref_counts) > 500, "the sequence must exercise pixels fused on other ranks"
    assert sum(c["conflict_count"] for c in ref_counts) > 0
    for r in range(G):              # counters after the collective export are the union's, on every rank
        assert out[r][2]["count"] == ref.shape[0]


@pytest.mark.gpu
def test_stream_shard_world1_equals_plain_path():
    over = dict(preprocess=1, stereo_border=10.0, max_sqrt_vertices=400)
    seq = _seq(10, noise_mm=8.0)
    plain = capi.SurfelMap(capi.make_config(**CAM, **over))
    sm = capi.SurfelMap(capi.make_config(**CAM, **over))
    mp = sharded.StreamShard(sm, 0, 1)
    for fr in seq:
        plain.process_frame(*fr)
        c = mp.process_frame(*fr)
        p = plain.counts()
        assert all(c[k] == p[k] for k in KEYS), (c, p)
    a, b = mp.export_dense(), plain.download_model()
    assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.gpu
def test_stream_shard_rccl_world1():
    """the production binding: ncclAllReduce on the context's stream, bound at run time; one rank"""
    over = dict(preprocess=0, stereo_border=10.0, max_sqrt_vertices=400, fuse_thresh=0.05)
    seq = _seq(8)
    ref, ref_counts = _oracle(seq, over)
    sm = capi.SurfelMap(capi.make_config(**CAM, **over, compact_period=3))
    mp = sharded.StreamShard(sm, 0, 1, "rccl", capi.rccl_unique_id())
    for fr, b in zip(seq, ref_counts):
        a = mp.process_frame(*fr)
        assert all(a[k] == b[k] for k in KEYS), (a, b)
    got = mp.export_dense()
    sm.shard_rccl_finalize()
    assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def _cap_frames():
    """frames whose model holds more surfels in view than pixels, then frames far behind all of them: every surfel in view
    conflicts, the W*H cap (src/GlobalModel.cpp:54-57) binds -- in slot order over ALL ranks"""
    cam = dict(width=64, height=48, fx=50.0, fy=50.0, cx=31.5, cy=23.5)
    rng = np.random.default_rng(5)
    W, H = cam["width"], cam["height"]
    rgb = rng.integers(0, 255, (H, W, 3), dtype=np.uint8)
    sem = np.full((H, W), 3, np.uint8)
    pose = np.eye(4, dtype=np.float32).T.reshape(16).copy()
    near = np.full((H, W), 4000, np.uint16)
    # every frame a little closer than the last, so nothing conflicts or fuses: six layers of W*H/2 surfels
    frames = [(rgb, near - np.uint16(40 * k), sem, pose) for k in range(7)]
    far = np.full((H, W), 20000, np.uint16)
    mid = np.full((H, W), 3900, np.uint16)
    frames += [(rgb, far, sem, pose), (rgb, mid, sem, pose), (rgb, far, sem, pose), (rgb, far, sem, pose)]
    return cam, frames


@pytest.mark.gpu
@pytest.mark.parametrize("G,period", [(1, 1000), (2, 1000), (3, 2), (4, 1000)])
def test_stream_shard_conflict_cap_is_applied_in_global_slot_order(G, period):
    """more conflicts than W*H over all ranks: the first W*H in the slot order of the single-GPU model take effect, whichever
    rank owns them (k_shard_cap_pack / k_shard_cap_repair) -- counters every frame and the union equal the oracle's"""
    cam, frames = _cap_frames()
    over = dict(preprocess=0, stereo_border=0.0, max_sqrt_vertices=300, conflict_cap=1)
    o = ol.Oracle(ol.make_config(**cam, **over))
    ref_counts = []
    for fr in frames:
        o.process_frame(*fr)
        ref_counts.append(o.counts())
    ref = o.download_model()
    P = cam["width"] * cam["height"]
    assert sum(c["conflict_count"] == P for c in ref_counts) >= 2, [c["conflict_count"] for c in ref_counts]     # the cap binds
    out = _run_threads(G, frames, over, period, cam)
    for r in range(G):
        for f, (a, b) in enumerate(zip(out[r][1], ref_counts)):
            assert all(a[k] == b[k] for k in KEYS), (r, f, {k: (a[k], b[k]) for k in KEYS})
    union = sharded.StreamShard.union([x[0] for x in out])
    assert union.shape == ref.shape and np.array_equal(union.view(np.uint32), ref.view(np.uint32))


def _run_threads_script(G, seq, over, period, script):
    """like _run_threads, but every rank runs `script(mp, sm, seq)` (the same call sequence on all ranks)"""
    grp = sharded.ThreadGroup(G)
    out, errs = [None] * G, []

    def work(r):
        try:
            sm = capi.SurfelMap(capi.make_config(**CAM, **over, compact_period=period))
            mp = sharded.StreamShard(sm, r, G, sharded.ThreadCollective(grp, r, sm) if G > 1 else None)
            out[r] = script(mp, sm, seq)
            sm.close()
        except Exception as e:
            errs.append((r, repr(e)))
            grp.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errs, errs
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("G", [2, 3])
def test_stream_shards_export_midway_then_continue(G):
    """a collective export (it compacts) in the middle of the stream, an explicit sm_shard_compact, the depth filter chain on"""
    over = dict(preprocess=1, stereo_border=10.0, max_sqrt_vertices=400, fuse_thresh=0.05)
    seq = _seq(14, noise_mm=2.0, seed=23)
    o = ol.Oracle(ol.make_config(**CAM, **over))
    mids = {}
    for k, fr in enumerate(seq):
        o.process_frame(*fr)
        if k == 6:
            mids["model"] = o.download_model()
    ref = o.download_model()

    def script(mp, sm, seq):
        for k, fr in enumerate(seq):
            mp.process_frame(*fr)
            if k == 6:
                mid = mp.export_dense()
            if k == 9:
                sm.shard_compact()
        return mid, mp.export_dense(), sm.counts()

    out = _run_threads_script(G, seq, over, 5, script)
    mid = sharded.StreamShard.union([x[0] for x in out])
    fin = sharded.StreamShard.union([x[1] for x in out])
    assert mid.shape == mids["model"].shape and np.array_equal(mid.view(np.uint32), mids["model"].view(np.uint32))
    assert fin.shape == ref.shape and np.array_equal(fin.view(np.uint32), ref.view(np.uint32))
    oc = o.counts()
    for x in out:
        assert all(x[2][k] == oc[k] for k in KEYS), (x[2], oc)


@pytest.mark.gpu
def test_stream_shards_under_capacity_pressure():
    """slots (live + dead) approach MAX_VERTICES: the schedule must compact early, identically on every rank, and the result
    stays the oracle's as long as the surfels themselves fit"""
    over = dict(preprocess=0, stereo_border=10.0, fuse_thresh=0.05)
    seq = _seq(14, noise_mm=2.0, seed=29)
    o = ol.Oracle(ol.make_config(**CAM, **over, max_sqrt_vertices=400))
    for fr in seq:
        o.process_frame(*fr)
    ref = o.download_model()
    n = ref.shape[0]
    side = int(np.ceil(np.sqrt(n + CAM["width"] * CAM["height"] // 2 + 2048)))        # room for the model + one frame of candidates, little more
    o2 = ol.Oracle(ol.make_config(**CAM, **over, max_sqrt_vertices=side))
    for fr in seq:
        o2.process_frame(*fr)
    assert np.array_equal(o2.download_model().view(np.uint32), ref.view(np.uint32))

    def script(mp, sm, seq):
        for fr in seq:
            mp.process_frame(*fr)
        return mp.export_dense(), sm.counts()

    out = _run_threads_script(2, seq, dict(over, max_sqrt_vertices=side), 1000, script)
    fin = sharded.StreamShard.union([x[0] for x in out])
    assert fin.shape == ref.shape and np.array_equal(fin.view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_stream_shards_kitti_size():
    """BASELINE configs[3]'s image size (1242x375): one stream over two ranks (two HIP contexts on the one GPU), six frames
    with fuses and conflicts, one compaction between frames; counters every frame and the union against the oracle"""
    cam = dict(synth.KITTI)
    over = dict(preprocess=0, fuse_thresh=0.05, max_sqrt_vertices=1500)
    seq = synth.make_sequence(cam, synth.kitti_trajectory(6), seed=5, noise_mm=4.0)
    o = ol.Oracle(ol.make_config(**cam, **over))
    ref_counts = []
    for fr in seq:
        o.process_frame(*fr)
        ref_counts.append(o.counts())
    ref = o.download_model()
    assert sum(c["fused_count"] for c in ref_counts) > 50000
    grp = sharded.ThreadGroup(2)
    out, errs = [None, None], []

    def work(r):
        try:
            sm = capi.SurfelMap(capi.make_config(**cam, **over, compact_period=4))
            mp = sharded.StreamShard(sm, r, 2, sharded.ThreadCollective(grp, r, sm))
            cs = [mp.process_frame(*fr) for fr in seq]
            out[r] = (mp.export_dense(), cs)
            sm.close()
        except Exception as e:
            errs.append((r, repr(e)))
            grp.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errs, errs
    for r in range(2):
        for f, (a, b) in enumerate(zip(out[r][1], ref_counts)):
            assert all(a[k] == b[k] for k in KEYS), (r, f, {k: (a[k], b[k]) for k in KEYS})
    union = sharded.StreamShard.union([x[0] for x in out])
    assert union.shape == ref.shape and np.array_equal(union.view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_stream_shards_random_configurations(seed):
    """seeded random image sizes, thresholds, noise, compaction periods, rank counts, filter chain on / off, one export in the
    middle: the union and the counters of every frame against the oracle"""
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(96, 260)), int(rng.integers(48, 130))
    cam = dict(width=W, height=H, fx=float(rng.uniform(80, 160)), fy=float(rng.uniform(80, 160)), cx=W / 2 - 0.5, cy=H / 2 - 0.5)
    G = int(rng.integers(1, 5))
    over = dict(preprocess=int(rng.integers(0, 2)), stereo_border=float(rng.choice([0.0, 8.0, 20.0])), max_sqrt_vertices=420,
                fuse_thresh=float(rng.choice([0.0, 0.03, 0.08])))
    period = int(rng.choice([1, 2, 3, 5, 8, 1000]))
    n = int(rng.integers(6, 13))
    seq = synth.make_sequence(cam, synth.kitti_trajectory(n), seed=int(rng.integers(1, 1000)), noise_mm=float(rng.choice([0.0, 2.0, 6.0])))
    mid_at = int(rng.integers(1, n - 1))
    o = ol.Oracle(ol.make_config(**cam, **over))
    ref_counts, mid_ref = [], None
    for k, fr in enumerate(seq):
        o.process_frame(*fr)
        ref_counts.append(o.counts())
        if k == mid_at:
            mid_ref = o.download_model()
    ref = o.download_model()
    grp = sharded.ThreadGroup(G)
    out, errs = [None] * G, []

    def work(r):
        try:
            sm = capi.SurfelMap(capi.make_config(**cam, **over, compact_period=period))
            mp = sharded.StreamShard(sm, r, G, sharded.ThreadCollective(grp, r, sm) if G > 1 else None)
            cs, mid = [], None
            for k, fr in enumerate(seq):
                cs.append(mp.process_frame(*fr))
                if k == mid_at:
                    mid = mp.export_dense()
            out[r] = (mp.export_dense(), cs, mid)
            sm.close()
        except Exception as e:
            errs.append((r, repr(e)))
            grp.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errs, (errs, cam, G, over, period)
    for r in range(G):
        for f, (a, b) in enumerate(zip(out[r][1], ref_counts)):
            assert all(a[k] == b[k] for k in KEYS), (seed, r, f, {k: (a[k], b[k]) for k in KEYS}, cam, G, over, period)
    mid = sharded.StreamShard.union([x[2] for x in out])
    fin = sharded.StreamShard.union([x[0] for x in out])
    assert mid.shape == mid_ref.shape and np.array_equal(mid.view(np.uint32), mid_ref.view(np.uint32)), (seed, "mid")
    assert fin.shape == ref.shape and np.array_equal(fin.view(np.uint32), ref.view(np.uint32)), (seed, "end")
