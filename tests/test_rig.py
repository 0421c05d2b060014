"""BASELINE configs[4]: a G-camera rig, one camera per rank, consolidated into a single GlobalModel
(surfelmapping_amd/dist.py::RigMapper).

Definition under test (reference operations only): the rank slices concatenated in rank order, then
SurfelMapping::cleanPoints (src/SurfelMapping.cpp:496-532) of that union against every camera's latest view, in rank
order.  `definition()` evaluates exactly that on ONE oracle instance; the distributed form (every rank cleans its own
slice against all views, then the slices are gathered) must give the same surfels bit for bit:
  * CPU, G = 2, 3 threads and 2 gloo processes with oracle-backed ranks (the host logic: view all-gather, id-0 exemption
    on the first non-empty slice only, conflict totals, gather order);
  * GPU, G = 2 and 4 HIP contexts on the one GPU (ThreadComm), the product path end to end."""
import ctypes as C
import math
import os
import socket
import sys
import threading

import numpy as np
import pytest

import oracle_lib as ol
from surfelmapping_amd import dist as smd
from surfelmapping_amd import sharded, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAM = dict(width=192, height=80, fx=110.0, fy=110.0, cx=95.5, cy=39.5)
OVER = dict(preprocess=0, stereo_border=12.0, max_sqrt_vertices=500)
N_FRAMES = 5


def rank_stream(rank, world, cam=None, n_frames=None):
    """camera `rank` of the rig: same forward motion, yawed by rank * 12 degrees (overlapping fields of view, so that the
    cameras see each other's surfels).  Every camera looks at its own set of parked cars (boxes): a car only one camera has
    seen is contradicted by the others' depth -- the situation the cross-camera conflict pass is for."""
    poses = [synth.pose_matrix(0.0, 0.0, 0.6 * k, 12.0 * rank + 0.4 * math.sin(k)) for k in range(n_frames or N_FRAMES)]
    return synth.make_sequence(cam or CAM, poses, seed=41, noise_mm=4.0 + rank, scene=synth.Scene(41 + rank, n_boxes=12, length=22.0))


class OracleRigBackend:
    """oracle-backed rank (CPU tests): cleanPoints with the id-0 exemption switched per slice"""

    def __init__(self):
        self.o = ol.Oracle(ol.make_config(**CAM, **OVER))
        self.L = ol.lib()
        self.L.smo_set_exempt_id.argtypes = [C.c_void_p, C.c_int32]
        self.L.smo_set_conflict_limit.argtypes = [C.c_void_p, C.c_int64]
        self.L.smo_count_clean_conflicts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]

    def process_frame(self, *fr):
        return self.o.process_frame(*fr)

    def counts(self):
        return self.o.counts()

    def download_model(self):
        return self.o.download_model()

    def clean_points_slice(self, depth, sem, pose, exempt_first, cap_hook=None):
        self.L.smo_set_exempt_id(self.o._h, 0 if exempt_first else -1)
        if cap_hook is not None:
            d = np.ascontiguousarray(depth, np.uint16); s_ = np.ascontiguousarray(sem, np.uint8); p = np.ascontiguousarray(pose, np.float32)
            n = C.c_uint32()
            assert self.L.smo_count_clean_conflicts(self.o._h, d.ctypes.data, s_.ctypes.data, p.ctypes.data, C.byref(n)) == 0
            self.L.smo_set_conflict_limit(self.o._h, int(cap_hook(n.value)))
        self.o.clean_points(depth, sem, pose)
        self.L.smo_set_conflict_limit(self.o._h, -1)
        self.L.smo_set_exempt_id(self.o._h, 0)


def definition(world, cam=None, over=None, n_frames=None):
    """the single GlobalModel by its definition, on one oracle: union in rank order, cleanPoints per view in rank order"""
    slices, views = [], []
    CAM_, OVER_, NF = cam or CAM, over or OVER, n_frames or N_FRAMES
    for r in range(world):
        o = ol.Oracle(ol.make_config(**CAM_, **OVER_))
        seq = rank_stream(r, world, CAM_, NF)
        for fr in seq:
            o.process_frame(*fr)
        slices.append(o.download_model())
        views.append(seq[-1][1:])
    g = ol.Oracle(ol.make_config(**CAM_, **dict(OVER_, max_sqrt_vertices=int(math.ceil(math.sqrt(sum(x.shape[0] for x in slices)))) + 8)))
    g.upload_model(np.concatenate(slices, axis=0))
    g.set_tick(NF)
    conflicts = []
    for depth, sem, pose in views:
        g.clean_points(depth, sem, pose)
        conflicts.append(g.counts()["conflict_count"])
    return g.download_model(), [s.shape[0] for s in slices], conflicts


def run_threads(world, make_backend, cam=None, n_frames=None):
    group = sharded.ThreadGroup(world)
    out, err = [None] * world, []
    CAM_ = cam or CAM

    def work(r):
        try:
            mp = smd.RigMapper(make_backend(r), sharded.ThreadComm(group, r), CAM_["width"] * CAM_["height"])
            for fr in rank_stream(r, world, CAM_, n_frames):
                mp.process_frame(*fr)
            out[r] = mp.consolidate()
        except BaseException as e:          # a failing rank must not leave the others waiting at the barrier
            err.append(e)
            group.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    if err:
        raise err[0]
    return out


def check(out, world, cam=None, over=None, n_frames=None):
    model, sizes, conflicts = definition(world, cam, over, n_frames)
    assert sum(conflicts) > 50, conflicts                      # the views do clean each other's surfels
    assert model.shape[0] < sum(sizes)
    for r in range(world):
        got, counts, per_view = out[r]
        assert per_view == conflicts
        assert sum(counts) == model.shape[0]
        assert np.array_equal(got.view(np.uint32), model.view(np.uint32)), f"rank {r}"


@pytest.mark.parametrize("world", [2, 3])
def test_rig_consolidation_oracle_ranks_in_threads(world):
    check(run_threads(world, lambda r: OracleRigBackend()), world)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_rig_consolidation_hip_contexts_equal_the_definition(world):
    from surfelmapping_amd import capi
    check(run_threads(world, lambda r: capi.SurfelMap(capi.make_config(**CAM, **OVER))), world)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rig_two_cameras_at_1920x1080():
    """the image size of BASELINE configs[4] (8 x 1920x1080, one camera per GPU), two of the cameras as two HIP contexts on
    the one GPU: three frames each, then the consolidation into a single GlobalModel, against the one-oracle definition"""
    from surfelmapping_amd import capi
    cam = dict(synth.HD)
    over = dict(preprocess=0, stereo_border=12.0, max_sqrt_vertices=2200)
    check(run_threads(2, lambda r: capi.SurfelMap(capi.make_config(**cam, **over)), cam, 3), 2, cam, over, 3)


def run_threads_native(world, cam=None, over=None, n_frames=None, collective="threads"):
    """the consolidation inside the HIP core (sm_rig_consolidate): G contexts on the one GPU, ranks = threads"""
    from surfelmapping_amd import capi
    group = sharded.ThreadGroup(world)
    out, err = [None] * world, []
    CAM_, OVER_ = cam or CAM, over or OVER

    def work(r):
        try:
            sm = capi.SurfelMap(capi.make_config(**CAM_, **OVER_))
            glob = capi.SurfelMap(capi.make_config(**CAM_, **dict(OVER_, max_sqrt_vertices=int(OVER_["max_sqrt_vertices"] * math.sqrt(world)) + 8)))
            mp = smd.RigMapper(sm, sharded.ThreadComm(group, r), CAM_["width"] * CAM_["height"])
            if collective == "rccl":
                mp.enable_native("rccl", capi.rccl_unique_id())
            else:
                mp.enable_native(sharded.ThreadCollective(group, r, sm) if world > 1 else None)
            for fr in rank_stream(r, world, CAM_, n_frames):
                mp.process_frame(*fr)
            total, per_view = mp.consolidate_native(glob)
            out[r] = (glob.download_model(), total, per_view)
            if collective == "rccl":
                sm.shard_rccl_finalize()
        except BaseException as e:
            err.append(e)
            group.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    if err:
        raise err[0]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_rig_consolidation_inside_the_core(world):
    """sm_rig_consolidate: the whole consolidation in the HIP core, the ranks' data crossing through the installed
    collective; every rank ends with the single GlobalModel of the one-oracle definition in its second context"""
    model, sizes, conflicts = definition(world)
    out = run_threads_native(world)
    for r in range(world):
        got, total, per_view = out[r]
        assert per_view == conflicts and total == model.shape[0]
        assert np.array_equal(got.view(np.uint32), model.view(np.uint32)), f"rank {r}"


@pytest.mark.gpu
def test_rig_consolidation_inside_the_core_rccl_world1():
    """the production binding (RCCL on the core's stream), one rank: the model cleaned against its own last view"""
    model, sizes, conflicts = definition(1)
    out = run_threads_native(1, collective="rccl")
    got, total, per_view = out[0]
    assert per_view == conflicts and total == model.shape[0]
    assert np.array_equal(got.view(np.uint32), model.view(np.uint32))


# ---------------------------------------------------------------- 2 gloo processes (torch.distributed on host arrays)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        import test_rig as tr
        mp = smd.RigMapper(tr.OracleRigBackend(), sharded.TorchComm(), CAM["width"] * CAM["height"])
        for fr in tr.rank_stream(rank, world):
            mp.process_frame(*fr)
        model, counts, per_view = mp.consolidate()
        q.put((rank, counts, per_view, model.view(np.uint32).tobytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rig_consolidation_two_gloo_processes():
    import torch.multiprocessing as tmp
    world = 2
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    model, sizes, conflicts = definition(world)
    for rank, counts, per_view, blob in res:
        assert per_view == conflicts and sum(counts) == model.shape[0]
        assert np.array_equal(np.frombuffer(blob, np.uint32).reshape(-1, 12), model.view(np.uint32))


# ---------------------------------------------------------------- the W*H conflict cap across slices (src/GlobalModel.cpp:54-57)
CAP_CAM = dict(width=48, height=32, fx=40.0, fy=40.0, cx=23.5, cy=15.5)
CAP_OVER = dict(preprocess=0, stereo_border=0.0, max_sqrt_vertices=120, conflict_cap=1)
CAP_SIZES = (1000, 2500, 3000)              # slice 0 leaves 536 of the 1 536 records to slice 1, slice 2 gets none (first view)


def cap_slice(rank):
    rng = np.random.default_rng(100 + rank)
    n = CAP_SIZES[rank]
    m = synth.seeded_model(n, tick=1, seed=50 + rank)
    m[:, 0] = rng.uniform(-1.5, 1.5, n)
    m[:, 1] = rng.uniform(-1.0, 1.0, n)
    m[:, 2] = rng.uniform(3.0, 6.0, n)
    m[:, 3] = rng.uniform(0.5, 3.5, n).astype(np.float32)
    return m


def cap_view(rank):
    """every camera sees a wall at 12 + rank metres from the origin: all surfels in view are contradicted (clean mode: far - 15 = 15 m)"""
    return (np.full((32, 48), 12000 + 1000 * rank, np.uint16), np.zeros((32, 48), np.uint8),
            np.eye(4, dtype=np.float32).T.reshape(16).copy())


def cap_definition(world):
    g = ol.Oracle(ol.make_config(**CAP_CAM, **CAP_OVER))
    g.upload_model(np.concatenate([cap_slice(r) for r in range(world)], axis=0))
    conflicts = []
    for r in range(world):
        g.clean_points(*cap_view(r))
        conflicts.append(g.counts()["conflict_count"])
    return g.download_model(), conflicts


def run_cap_threads(world, make_backend, native=False):
    group = sharded.ThreadGroup(world)
    out, err = [None] * world, []

    def work(r):
        try:
            be = make_backend(r)
            be.upload_model(cap_slice(r))
            mp = smd.RigMapper(be, sharded.ThreadComm(group, r), CAP_CAM["width"] * CAP_CAM["height"])
            mp.last = cap_view(r)
            if native:
                from surfelmapping_amd import capi
                glob = capi.SurfelMap(capi.make_config(**CAP_CAM, **CAP_OVER))
                mp.enable_native(sharded.ThreadCollective(group, r, be))
                total, per_view = mp.consolidate_native(glob)
                out[r] = (glob.download_model(), total, per_view)
            else:
                model, counts, per_view = mp.consolidate()
                out[r] = (model, sum(counts), per_view)
        except BaseException as e:
            err.append(e)
            group.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    if err:
        raise err[0]
    return out


def check_cap(out, world):
    model, conflicts = cap_definition(world)
    P = CAP_CAM["width"] * CAP_CAM["height"]
    assert conflicts[0] == P and max(conflicts) == P, conflicts         # the cap binds
    for r in range(world):
        got, total, per_view = out[r]
        assert per_view == conflicts and total == model.shape[0], (per_view, conflicts, total, model.shape)
        assert np.array_equal(got.view(np.uint32), model.view(np.uint32)), f"rank {r}"


class _OracleUploadBackend(OracleRigBackend):
    def __init__(self):
        self.o = ol.Oracle(ol.make_config(**CAP_CAM, **CAP_OVER))
        self.L = ol.lib()
        self.L.smo_set_exempt_id.argtypes = [C.c_void_p, C.c_int32]
        self.L.smo_set_conflict_limit.argtypes = [C.c_void_p, C.c_int64]
        self.L.smo_count_clean_conflicts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]

    def upload_model(self, m):
        self.o.upload_model(m)


def test_rig_conflict_cap_is_shared_exactly_oracle_ranks():
    """more conflicts than pixels in every view: the first W*H in the surfel order of the union take effect, i.e. slice 0's
    all, slice 1's first 536, none of slice 2's (first view) -- each rank applies its share, nobody refuses"""
    check_cap(run_cap_threads(3, lambda r: _OracleUploadBackend()), 3)


@pytest.mark.gpu
@pytest.mark.parametrize("native", [False, True])
def test_rig_conflict_cap_is_shared_exactly_hip_contexts(native):
    from surfelmapping_amd import capi
    check_cap(run_cap_threads(3, lambda r: capi.SurfelMap(capi.make_config(**CAP_CAM, **CAP_OVER)), native), 3)


# ---------------------------------------------------------------- the single GlobalModel DURING a run (sm_rig_consolidate_step)
def step_definition(world, steps, n_frames):
    """the incremental GlobalModel by its definition, on oracles: at every step the ranks' surfels created since the previous
    step (still alive, creation order) are appended in rank order (GlobalModel::concatenate), then the model is cleaned against
    every camera's latest view in rank order (SurfelMapping::cleanPoints)"""
    ranks = [ol.Oracle(ol.make_config(**CAM, **OVER)) for _ in range(world)]
    seqs = [rank_stream(r, world, CAM, n_frames) for r in range(world)]
    g = ol.Oracle(ol.make_config(**CAM, **dict(OVER, max_sqrt_vertices=int(OVER["max_sqrt_vertices"] * math.sqrt(world)) + 8)))
    out, last_t, k0 = [], -1.0e30, 0
    for k1 in steps:
        lists = []
        for r in range(world):
            for fr in seqs[r][k0:k1]:
                ranks[r].process_frame(*fr)
            m = ranks[r].download_model()
            lists.append(m[m[:, 6] > last_t])
        g.upload_model(np.concatenate([g.download_model()] + lists, axis=0))
        for r in range(world):
            g.clean_points(*seqs[r][k1 - 1][1:])
        out.append((g.download_model(), sum(x.shape[0] for x in lists)))
        last_t, k0 = float(k1 - 1), k1          # frame k carries the time stamp k: everything created so far is <= k1 - 1
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_rig_incremental_global_model_equals_the_definition(world):
    from surfelmapping_amd import capi
    steps, n_frames = (3, 5, 7), 7
    ref = step_definition(world, steps, n_frames)
    assert ref[-1][0].shape[0] > 5000 and all(n > 500 for _, n in ref)
    group = sharded.ThreadGroup(world)
    out, err = [None] * world, []

    def work(r):
        try:
            sm = capi.SurfelMap(capi.make_config(**CAM, **OVER))
            glob = capi.SurfelMap(capi.make_config(**CAM, **dict(OVER, max_sqrt_vertices=int(OVER["max_sqrt_vertices"] * math.sqrt(world)) + 8)))
            mp = smd.RigMapper(sm, sharded.ThreadComm(group, r), CAM["width"] * CAM["height"])
            mp.enable_native(sharded.ThreadCollective(group, r, sm) if world > 1 else None)
            seq = rank_stream(r, world, CAM, n_frames)
            got, k0 = [], 0
            for k1 in steps:
                for fr in seq[k0:k1]:
                    mp.process_frame(*fr)
                new, tot = mp.consolidate_step_native(glob)
                got.append((glob.download_model(), new, tot))
                k0 = k1
            own = sm.download_model()
            out[r] = (got, own)
        except BaseException as e:
            err.append(e)
            group.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    if err:
        raise err[0]
    for r in range(world):
        for k, ((model, new, tot), (want, want_new)) in enumerate(zip(out[r][0], ref)):
            assert new == want_new and tot == want.shape[0], (r, k, new, want_new, tot, want.shape)
            assert np.array_equal(model.view(np.uint32), want.view(np.uint32)), f"rank {r} step {k}"
        # the camera's own model is what it would be alone
        o = ol.Oracle(ol.make_config(**CAM, **OVER))
        for fr in rank_stream(r, world, CAM, n_frames):
            o.process_frame(*fr)
        assert np.array_equal(out[r][1].view(np.uint32), o.download_model().view(np.uint32))
