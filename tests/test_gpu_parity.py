"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same
seeded inputs.  Bar: surfel count exact and every stored field bit-exact (both sides evaluate
the same IEEE-fp32 expression trees; DESIGN.md "Arithmetic")."""
import math
import os

import numpy as np
import pytest

import oracle_lib as ol
from backends import assert_models_equal, make
from surfelmapping_amd import capi, synth

pytestmark = pytest.mark.gpu

SMALL = dict(width=320, height=120, fx=180.0, fy=180.0, cx=159.5, cy=59.5)
IDENT = np.eye(4, dtype=np.float32).T.reshape(16).copy()
COUNT_KEYS = ("count", "offset", "data_count", "conflict_count", "unstable_count", "fused_count",
              "visible_count", "tick")


def pair(cam, **over):
    args = (cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    over.setdefault("preprocess", 0)
    o = make("oracle", *args, **over)
    h = make("hip", *args, **over)
    return o, h


def check(o, h, what, counts=True, model=True):
    if counts:
        co, ch = o.counts(), h.counts()
        assert {k: co[k] for k in COUNT_KEYS} == {k: ch[k] for k in COUNT_KEYS}, what
    if model:
        assert_models_equal(o.download_model(), h.download_model(), what)


def run_sequence(o, h, seq, every=1):
    for k, fr in enumerate(seq):
        o.process_frame(*fr)
        h.process_frame(*fr)
        if k % every == 0 or k == len(seq) - 1:
            check(o, h, f"frame {k}")


def test_moving_camera_small():
    seq = synth.make_sequence(SMALL, synth.kitti_trajectory(8), seed=3)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=600)
    run_sequence(o, h, seq)
    assert o.counts()["count"] > 10000 and o.counts()["conflict_count"] > 0


def test_static_camera_fuses():
    seq = synth.make_sequence(SMALL, [synth.pose_matrix(0, 0, 0)] * 5, seed=4)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=600)
    run_sequence(o, h, seq)
    assert o.counts()["fused_count"] > 1000


def test_fuse_thresh_with_noise_exercises_both_fuse_branches():
    poses = [synth.pose_matrix(0, 0, 0.05 * k, 0.2 * math.sin(k)) for k in range(8)]
    seq = synth.make_sequence(SMALL, poses, seed=5, noise_mm=4.0)
    o, h = pair(SMALL, stereo_border=20.0, fuse_thresh=0.05, max_sqrt_vertices=600)
    run_sequence(o, h, seq)
    c = o.counts()
    assert c["fused_count"] > 500 and c["unstable_count"] > 500 and c["conflict_count"] > 0
    m = o.download_model()
    assert (m[:, 3] > 1.0).sum() > 500          # confidence accumulated


def test_config1_vga_identity_three_calls():
    """BASELINE configs[0]: 640x480, identity pose: call 1 reference only, 2 all new, 3 fuse."""
    cam = synth.VGA
    seq = synth.make_sequence(cam, [synth.pose_matrix(0, 0, 0)] * 3, seed=1)
    o, h = pair(cam, max_sqrt_vertices=1000)
    run_sequence(o, h, seq)
    c = h.counts()
    assert c["tick"] == 3 and c["fused_count"] > 0


def test_kitti_full_size_sequence():
    """BASELINE configs[1] shape: 1242x375 KITTI intrinsics, moving camera."""
    cam = synth.KITTI
    seq = synth.make_sequence(cam, synth.kitti_trajectory(7), seed=1)
    o, h = pair(cam, max_sqrt_vertices=2000)
    run_sequence(o, h, seq, every=2)
    c = h.counts()
    assert c["count"] == c["offset"] + c["unstable_count"] and c["count"] > 300000


def test_index_map_stage_bit_exact():
    seq = synth.make_sequence(SMALL, synth.kitti_trajectory(4), seed=6)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=600)
    run_sequence(o, h, seq)
    pose = synth.pose_to_colmajor(synth.pose_matrix(0.1, 0.0, 3.0, 1.0))
    o.stage_predict_indices(pose, 5, 30.0, 200)
    h.stage_predict_indices(pose, 5, 30.0, 200)
    io, ih = o.download_index_map(), h.download_index_map()
    np.testing.assert_array_equal(io[0], ih[0])
    for a, b, name in zip(io[1:], ih[1:], ("vertConf", "colorTime", "normRad")):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32), err_msg=name)
    assert (io[0] > 0).sum() > 1000
    assert o.counts()["visible_count"] == h.counts()["visible_count"]


@pytest.mark.parametrize("cap", [1, 0])
def test_conflict_cap_first_P_conflicts_only(cap):
    """conflictVbo holds W*H records (src/GlobalModel.cpp:54-57): with more conflicts than pixels
    only the first P in surfel order take effect."""
    cam = dict(width=48, height=32, fx=40.0, fy=40.0, cx=23.5, cy=15.5)
    o, h = pair(cam, stereo_border=0.0, conflict_cap=cap, max_sqrt_vertices=200)
    rng = np.random.default_rng(7)
    n = 20000
    m = synth.seeded_model(n, tick=5, seed=9)
    m[:, 0] = rng.uniform(-1.5, 1.5, n)
    m[:, 1] = rng.uniform(-1.0, 1.0, n)
    m[:, 2] = rng.uniform(3.0, 6.0, n)
    m[::7, 3] = 0.0                      # dead-already surfels (conf <= 0)
    o.upload_model(m)
    h.upload_model(m)
    dm = np.full((32, 48), 20.0, np.float32)     # everything measured farther -> conflicts
    sem = np.zeros((32, 48), np.uint8)
    o.set_frame(None, dm, sem)
    h.set_frame(None, dm, sem)
    o.stage_process_conflict(IDENT, 1.0, 30.0, 0.0, 0)
    h.stage_process_conflict(IDENT, 1.0, 30.0, 0.0, 0)
    assert o.counts()["conflict_count"] == h.counts()["conflict_count"]
    if cap:
        assert o.counts()["conflict_count"] == 48 * 32
    else:
        assert o.counts()["conflict_count"] > 48 * 32
    o.stage_update_conflict(); o.stage_back_mapping(); o.stage_build_model_map()
    h.stage_update_conflict(); h.stage_back_mapping(); h.stage_build_model_map()
    check(o, h, f"cap={cap}", counts=False)
    co, ch = o.counts(), h.counts()
    assert co["count"] == ch["count"] and co["offset"] == ch["offset"]


def test_clean_points():
    seq = synth.make_sequence(SMALL, synth.kitti_trajectory(5), seed=8)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=600)
    run_sequence(o, h, seq[:4])
    rgb, d, s, p = seq[1]
    o.clean_points(d, s, p)
    h.clean_points(d, s, p)
    check(o, h, "after cleanPoints")
    run_sequence(o, h, seq[4:])


def test_async_device_resident_frames_match_sync_path():
    seq = synth.make_sequence(SMALL, synth.kitti_trajectory(6), seed=10)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=600)
    P = SMALL["width"] * SMALL["height"]
    bufs = []
    for rgb, d, s, p in seq:
        dr, dd, ds = h.device_alloc(P * 3), h.device_alloc(P * 2), h.device_alloc(P)
        h.device_upload(dr, rgb); h.device_upload(dd, d); h.device_upload(ds, s)
        bufs.append((dr, dd, ds, p))
    for fr in seq:
        o.process_frame(*fr)
    for dr, dd, ds, p in bufs:
        h.process_frame_device(dr, dd, ds, p)      # enqueue only, no host sync in between
    h.sync()
    check(o, h, "async")


def test_map_save_load_roundtrip(tmp_path):
    seq = synth.make_sequence(SMALL, synth.kitti_trajectory(3), seed=11)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=600)
    run_sequence(o, h, seq)
    path = str(tmp_path / "map.bin")
    h.save_map(path, 3, 9)
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw[:4], np.uint32)[0])
    assert n == o.counts()["count"] and len(raw) == 12 + n * 48
    assert tuple(np.frombuffer(raw[4:12], np.int32)) == (3, 9)
    assert_models_equal(np.frombuffer(raw[12:], np.float32).reshape(n, 12), o.download_model(), "file")
    _, h2 = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=600)
    assert h2.load_map(path) == (3, 9)
    assert_models_equal(h2.download_model(), o.download_model(), "reloaded")


def test_capacity_is_reported_not_corrupted():
    cam = dict(width=64, height=48, fx=50.0, fy=50.0, cx=31.5, cy=23.5)
    o, h = pair(cam, stereo_border=0.0, max_sqrt_vertices=30)      # 900 surfels
    rgb = np.zeros((48, 64, 3), np.uint8)
    d = np.full((48, 64), 4000, np.uint16)
    s = np.zeros((48, 64), np.uint8)
    o.process_frame(rgb, d, s, IDENT); h.process_frame(rgb, d, s, IDENT)
    rc_o = o.process_frame(rgb, d, s, IDENT, allow=(0, -2))
    rc_h = h.process_frame(rgb, d, s, IDENT, allow=(0, -2))
    assert rc_o == rc_h == capi.SM_E_CAPACITY
    co, ch = o.counts(), h.counts()
    assert co["count"] == ch["count"] == 0 and co["unstable_count"] == ch["unstable_count"] > 900


def test_seeded_two_million_surfels_hd():
    """BASELINE configs[2] shape at a size the oracle finishes in seconds: 1920x1080, pre-seeded
    model, dense-depth frames."""
    cam = synth.HD
    n = 2_000_000
    o, h = pair(cam, max_sqrt_vertices=2000, conflict_cap=0)
    m = synth.seeded_model(n, tick=300, seed=2)
    o.upload_model(m); h.upload_model(m)
    o.set_tick(300); h.set_tick(300)
    seq = synth.make_sequence(cam, synth.kitti_trajectory(3), seed=2)
    for k, fr in enumerate(seq[1:]):
        o.process_frame(*fr); h.process_frame(*fr)
        check(o, h, f"hd frame {k}", model=(k == 1))
    c = h.counts()
    assert c["conflict_count"] > 1000 and c["visible_count"] > 10000


# ---------------------------------------------------------------- pre-processing chain p0a..p0e
def moving_boxes_sequence(cam, n, seed, noise_mm=3.0, box_speed=1.5):
    """Frames in which the car-sized boxes drive away from the camera (exercises removeMovings)."""
    out = []
    poses = synth.kitti_trajectory(n)
    for k, p in enumerate(poses):
        sc = synth.Scene(seed, n_boxes=10, length=40.0)
        sc.boxes[:, [2, 5]] += box_speed * k
        rgb, d, s = sc.render(synth.Camera(**cam), p, noise_mm=noise_mm, noise_seed=seed * 1000 + k)
        out.append((rgb, d, s, synth.pose_to_colmajor(p)))
    return out


def check_depth_textures(o, h, what):
    for which, name in ((0, "DEPTH_METRIC"), (1, "DEPTH_FILTERED"), (2, "LAST")):
        a, b = o.download_depth(which), h.download_depth(which)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{what}: {name} differs in {(a != b).sum()} pixels"


def test_preprocess_chain_small():
    cam = dict(width=320, height=120, fx=180.0, fy=180.0, cx=159.5, cy=59.5)
    seq = moving_boxes_sequence(cam, 6, seed=31)
    o, h = pair(cam, preprocess=1, stereo_border=20.0, max_sqrt_vertices=600)
    removed = 0
    for k, fr in enumerate(seq):
        o.process_frame(*fr); h.process_frame(*fr)
        check_depth_textures(o, h, f"frame {k}")
        check(o, h, f"frame {k}")
        if k:
            removed += int(((o.download_depth(1) > 0) & (o.download_depth(0) == 0)).sum())
    assert removed > 0, "removeMovings never fired: the test does not exercise p0e"
    assert o.counts()["count"] > 5000


def test_preprocess_chain_kitti_size():
    cam = synth.KITTI
    seq = moving_boxes_sequence(cam, 4, seed=32, noise_mm=8.0)
    o, h = pair(cam, preprocess=1, max_sqrt_vertices=2000)
    for k, fr in enumerate(seq):
        o.process_frame(*fr); h.process_frame(*fr)
        check_depth_textures(o, h, f"frame {k}")
    check(o, h, "kitti preprocess")
    assert o.counts()["count"] > 100000


def test_async_frames_overlap_the_depth_filter_chain():
    """With preprocess=1 the asynchronous entry point runs frame f+1's pre-processing on a second stream while frame f
    is still associating (double-buffered frame planes); synchronous and asynchronous calls may be mixed."""
    cam = dict(width=320, height=120, fx=180.0, fy=180.0, cx=159.5, cy=59.5)
    seq = moving_boxes_sequence(cam, 14, seed=33)
    o, h = pair(cam, preprocess=1, stereo_border=20.0, max_sqrt_vertices=600)
    P = cam["width"] * cam["height"]
    bufs = []
    for rgb, d, s, p in seq:
        dr, dd, ds = h.device_alloc(P * 3), h.device_alloc(P * 2), h.device_alloc(P)
        h.device_upload(dr, rgb); h.device_upload(dd, d); h.device_upload(ds, s)
        bufs.append((dr, dd, ds, p))
    for k, fr in enumerate(seq):
        o.process_frame(*fr)
        if k in (4, 9):
            h.process_frame(*fr)                    # synchronous call in between: its chain runs on the main stream
            check_depth_textures(o, h, f"frame {k}")
        else:
            h.process_frame_device(*bufs[k])        # enqueue only
        if k in (6, 13):
            h.sync()
            check_depth_textures(o, h, f"frame {k}")
            check(o, h, f"frame {k}")
    assert o.counts()["count"] > 5000


@pytest.mark.parametrize("env", [{"SM_COMPACT_TICKETS": "1"}, {"SM_DEFER_ASSOC": "0"}, {"SM_PASS_TRACE": "@tmp"},
                                 {"SM_TWO_LAUNCH": "0"}, {"SM_PASS_SPLIT": "1"}])
def test_kernel_variants_behind_switches_stay_bit_exact(env):
    """The switches that are left after round 3's pruning (the rejected kernel variants are gone).  SM_COMPACT_TICKETS=1: the
    in-place compaction hands its moving tiles out from a ticket counter (the form used as soon as two contexts share a GPU: no
    co-residency assumption).  SM_DEFER_ASSOC=0: every asynchronous frame launches its own association instead of handing it
    to the next frame's preparation launch (k_assoc_prep).  SM_PASS_TRACE: the per-workgroup time stamps of tools/pass_trace.py
    must not change a result.  SM_TWO_LAUNCH=0: the fixup step (publisher, cap repair, candidate count) keeps its own launch
    between the pass and the association instead of riding on the next frame's preparation launch.  SM_PASS_SPLIT=1: k_surfel_pass
    takes whole tiles per workgroup on small models too (the default there is a quarter tile; large models use whole tiles
    anyway).  Each runs the deferred-compaction and fuzz tests in a child process."""
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tmp = None
    if env.get("SM_PASS_TRACE") == "@tmp":      # the per-workgroup time stamps (tools/pass_trace.py) must not change a result either
        tmp = tempfile.TemporaryDirectory()
        env = dict(env, SM_PASS_TRACE=os.path.join(tmp.name, "trace"))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "tests/test_deferred_compaction.py",
                        "tests/test_fuzz_gpu.py"], cwd=root, env=dict(os.environ, SM_FUZZ_SEEDS="24", **env),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-1500:]
    if tmp is not None:
        dumps = os.listdir(tmp.name)
        assert any(f.endswith(".bin") and os.path.getsize(os.path.join(tmp.name, f)) > 0 for f in dumps), dumps
        tmp.cleanup()


def assert_models_equal_nan_tolerant(a, b, what=""):
    """As assert_models_equal, but any NaN equals any NaN: the raw feedback cloud has no neighbour test
    (surfel_feedback.vert), so border pixels get 0/0 normals whose NaN payload/sign is not specified."""
    assert a.shape == b.shape, f"{what}: {a.shape} vs {b.shape}"
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), what
    au, bu = a.view(np.uint32).copy(), b.view(np.uint32).copy()
    au[na] = 0; bu[nb] = 0
    assert np.array_equal(au, bu), what


def test_reset_then_reinitialise_from_raw_cloud():
    """SurfelMapping::reset (src/SurfelMapping.cpp:436-441) + the tick==0 branch (:161-169)."""
    seq = synth.make_sequence(SMALL, synth.kitti_trajectory(7), seed=13)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=600)
    run_sequence(o, h, seq[:3])
    o.reset(); h.reset()
    assert o.counts()["count"] == h.counts()["count"] == 0 and h.counts()["tick"] == 0
    o.process_frame(*seq[3]); h.process_frame(*seq[3])
    co, ch = o.counts(), h.counts()
    assert co["count"] == ch["count"] > 5000 and ch["tick"] == 1
    assert_models_equal_nan_tolerant(o.download_model(), h.download_model(), "after re-initialisation")
    for k, fr in enumerate(seq[4:]):
        o.process_frame(*fr); h.process_frame(*fr)
        assert {x: o.counts()[x] for x in COUNT_KEYS} == {x: h.counts()[x] for x in COUNT_KEYS}
        assert_models_equal_nan_tolerant(o.download_model(), h.download_model(), f"frame {k} after reset")


def test_seeded_eight_million_surfels_in_place_cull_under_load():
    """The in-place compaction hand-off (tile flags) with ~8000 tiles in flight: 8 M seeded surfels,
    kills spread over the whole array, conflict cap off (as in the stress benchmark)."""
    cam = synth.HD
    n = 8_000_000
    o, h = pair(cam, max_sqrt_vertices=3200, conflict_cap=0)
    m = synth.seeded_model(n, tick=300, seed=5)
    o.upload_model(m); h.upload_model(m)
    o.set_tick(300); h.set_tick(300)
    seq = synth.make_sequence(cam, synth.kitti_trajectory(4), seed=5)
    for k, fr in enumerate(seq[1:]):
        o.process_frame(*fr); h.process_frame(*fr)
        check(o, h, f"8M frame {k}", model=(k == 2))
    assert h.counts()["offset"] < h.counts()["count"] and o.counts()["conflict_count"] > 10000


def test_tile_bounds_skip_tiles_without_changing_results():
    """Whole 1024-surfel tiles whose bounding box is outside the view are skipped by the conflict pass and the
    splat; the result must not depend on it (A/B against disable_tile_bounds=1 and against the oracle)."""
    poses = synth.kitti_trajectory(40, step=1.6)            # drive away from the early surfels
    seq = synth.make_sequence(SMALL, poses, seed=14)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=800)
    args = (SMALL["width"], SMALL["height"], SMALL["fx"], SMALL["fy"], SMALL["cx"], SMALL["cy"])
    h_off = make("hip", *args, preprocess=0, stereo_border=20.0, max_sqrt_vertices=800, disable_tile_bounds=1)
    for k, fr in enumerate(seq):
        o.process_frame(*fr); h.process_frame(*fr); h_off.process_frame(*fr)
        if k % 8 == 7 or k == len(seq) - 1:
            check(o, h, f"bounds on, frame {k}")
            check(o, h_off, f"bounds off, frame {k}")
    log_on, log_off = h.read_frame_log(), h_off.read_frame_log()
    assert log_on["n_conf_skipped"][-1] > 50_000 and log_on["n_splat_skipped"][-1] > 10_000, log_on[-3:]
    assert log_off["n_conf_skipped"].sum() == 0 and log_off["n_splat_skipped"].sum() == 0
    assert np.array_equal(log_on["visible_count"], log_off["visible_count"])


def test_tile_bounds_with_yaw_and_turning_back():
    """The camera turns around and looks at old parts of the map again: skipped tiles must come back."""
    import math
    poses = ([synth.pose_matrix(0, 0, 0.8 * k, 0.0) for k in range(12)] +
             [synth.pose_matrix(0, 0, 0.8 * 11, 15.0 * k) for k in range(1, 13)] +      # turn 180 degrees
             [synth.pose_matrix(0, 0, 0.8 * 11 - 0.8 * k, 180.0) for k in range(1, 8)])  # drive back
    seq = synth.make_sequence(SMALL, poses, seed=15)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=800)
    run_sequence(o, h, seq, every=4)
    log = h.read_frame_log()
    assert log["n_conf_skipped"].max() > 10_000


def test_tile_bounds_with_the_camera_inside_the_boxes():
    """The side planes of the view volumes are applied to tile boxes that straddle the camera plane too (sm_kernels.h
    plane_guard): a linear form that is positive at all 8 corners is positive on the whole box.  Uploaded model scattered
    all around a camera that sits INSIDE every tile's box, with surfels within centimetres of the camera plane and close
    to the image borders there (where the 2-pixel margin times z shrinks to the rounding error the guard has to cover);
    small random pose changes; A/B against disable_tile_bounds=1 and against the oracle."""
    rng = np.random.default_rng(77)
    n_tiles, T = 48, 1024                                         # one compact cluster per 1024-surfel tile: a box of its own
    n = n_tiles * T
    m = np.zeros((n, 12), dtype=np.float32)
    for t in range(n_tiles):
        kind = t % 3
        if kind == 0:      # beside the camera, straddling its plane: everything of it in front is outside the image
            c = np.array([rng.choice([-1, 1]) * rng.uniform(1.5, 8.0), rng.uniform(-2, 2), rng.uniform(-1.0, 1.0)])
            hs = rng.uniform(0.3, 1.2, 3)
        elif kind == 1:    # tiny, on the ray of an image border, centimetres in front of / behind the camera plane
            z0 = rng.uniform(0.01, 0.12)
            u = rng.choice([0.0, float(SMALL["width"]), rng.uniform(0, SMALL["width"])])
            v = 0.0 if u not in (0.0, float(SMALL["width"])) and rng.random() < 0.5 else rng.uniform(0, SMALL["height"])
            c = np.array([(u - SMALL["cx"]) / SMALL["fx"] * z0, (v - SMALL["cy"]) / SMALL["fy"] * z0, z0])
            hs = rng.uniform(0.005, 0.06, 3)
        else:              # anywhere around, any size: some in view, some behind, some containing the camera
            c = np.array([rng.uniform(-8, 8), rng.uniform(-3, 3), rng.uniform(-4, 12)])
            hs = rng.uniform(0.1, 4.0, 3)
        m[t * T:(t + 1) * T, 0:3] = (c + rng.uniform(-1, 1, (T, 3)) * hs).astype(np.float32)
    m[:, 3] = rng.choice(np.array([0.9, 1.8, 2.7], dtype=np.float32), n)
    m[:, 4] = ((rng.integers(0, 19, n, dtype=np.uint32) << 24) | rng.integers(0, 1 << 24, n, dtype=np.uint32)).view(np.float32)
    m[:, 6] = 1.0; m[:, 7] = 1.0
    nv = rng.normal(size=(n, 3)); nv /= np.linalg.norm(nv, axis=1, keepdims=True)
    m[:, 8:11] = nv.astype(np.float32)
    m[:, 11] = rng.uniform(0.02, 0.1, n).astype(np.float32)
    poses = [synth.pose_matrix(0.002 * k, 0.001 * k, 0.003 * k, 0.4 * k) for k in range(6)]
    seq = synth.make_sequence(SMALL, poses, seed=21, noise_mm=5.0)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=500)
    args = (SMALL["width"], SMALL["height"], SMALL["fx"], SMALL["fy"], SMALL["cx"], SMALL["cy"])
    h_off = make("hip", *args, preprocess=0, stereo_border=20.0, max_sqrt_vertices=500, disable_tile_bounds=1)
    o.process_frame(*seq[0]); h.process_frame(*seq[0]); h_off.process_frame(*seq[0])
    for b in (o, h, h_off):
        b.upload_model(m)
    for k, fr in enumerate(seq[1:]):
        o.process_frame(*fr); h.process_frame(*fr); h_off.process_frame(*fr)
        check(o, h, f"bounds on, frame {k}")
        check(o, h_off, f"bounds off, frame {k}")
    log_on, log_off = h.read_frame_log(), h_off.read_frame_log()
    assert np.array_equal(log_on["visible_count"], log_off["visible_count"])
    assert np.array_equal(log_on["conflict_count"], log_off["conflict_count"])
    assert log_on["n_conf_skipped"].min() >= 8 * 1024 and log_on["n_splat_skipped"].min() >= 2 * 1024, log_on[-3:]     # (tiles really are skipped)
    assert log_on["visible_count"].min() > 2000 and log_on["conflict_count"].max() > 500, log_on[-3:]


def test_raw_feedback_cloud_matches_oracle():
    """FeedbackBuffer "RAW" (src/FeedbackBuffer.cpp:85-145, surfel_feedback.vert): the raw camera-frame cloud of the last
    frame through sm_download_raw_cloud -- empty before the second call (the reference computes it from the second
    processFrame on), then bit-identical to the oracle's restatement, also with the depth filter chain and after reset()."""
    for pre in (0, 1):
        seq = synth.make_sequence(SMALL, synth.kitti_trajectory(5), seed=17, noise_mm=3.0)
        o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=600, preprocess=pre)
        o.process_frame(*seq[0]); h.process_frame(*seq[0])
        assert h.download_raw_cloud().shape[0] == 0
        for k, fr in enumerate(seq[1:4]):
            o.process_frame(*fr); h.process_frame(*fr)
            a, b = o.download_raw_cloud(), h.download_raw_cloud()
            assert a.shape == b.shape and a.shape[0] > 2000, (pre, k, a.shape, b.shape)
            assert_models_equal_nan_tolerant(a, b, f"raw cloud pre={pre} frame {k}")
            assert np.all(b[:, 3] == np.float32(0.9)) and np.all(b[:, 6] == k + 1) and np.all(b[:, 5] == 0)
        o.reset(); h.reset()
        o.process_frame(*seq[4]); h.process_frame(*seq[4])
        assert_models_equal_nan_tolerant(o.download_raw_cloud(), h.download_raw_cloud(), f"raw cloud after reset pre={pre}")
        assert o.counts() == h.counts()
        assert_models_equal_nan_tolerant(o.download_model(), h.download_model(), f"model after raw-cloud downloads pre={pre}")


def test_async_host_buffers_match_oracle():
    """sm_process_frame_async: host images, no host wait -- the copy of frame f+1 overlaps frame f on a copy stream (three
    device input sets).  Pinned frame blocks and separate pinned buffers of the library (sm_host_alloc_frame / sm_host_alloc) and
    pageable arrays (staged inside the call), a null
    depth / semantic (keeps the previous texture, src/SurfelMapping.cpp:124-128), with the depth filter chain and without."""
    for pre in (0, 1):
        seq = moving_boxes_sequence(SMALL, 14, seed=8) if pre else synth.make_sequence(SMALL, synth.kitti_trajectory(14), seed=8, noise_mm=3.0)
        o, h = pair(SMALL, preprocess=pre, stereo_border=20.0, max_sqrt_vertices=700, fuse_thresh=0.03, compact_period=4)
        # a reader's three pinned buffer sets: two as frame blocks (one transfer per frame), one as three separate buffers
        ring = [h.host_frame(), tuple(h.host_array(x.shape, x.dtype) for x in seq[0][:3]), h.host_frame()]
        for k, (rgb, d, s_, p) in enumerate(seq):
            dd, ss = (None, None) if k == 6 else (d, s_)          # frame 6: rgb only
            o.process_frame(rgb, seq[k - 1][1] if k == 6 else d, seq[k - 1][2] if k == 6 else s_, p)
            if k % 2 == 0:                                       # every other frame from pinned memory, reused three frames later
                bufs = ring[(k // 2) % 3]
                if k >= 6:
                    h.inputs_consumed()
                for dst, src in zip(bufs, (rgb, d, s_)):
                    np.copyto(dst, src.reshape(dst.shape))
                h.process_frame_async(bufs[0], None if dd is None else bufs[1], None if ss is None else bufs[2], p)
            else:
                h.process_frame_async(np.ascontiguousarray(rgb), dd, ss, p)
            if k == 9:
                h.inputs_consumed()
        h.sync()
        check(o, h, f"async host buffers, preprocess={pre}")
