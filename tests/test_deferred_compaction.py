"""Deferred compaction (DESIGN.md "Deferred compaction"): culled surfels keep their slots, marked dead, until
the next compacting cull (every `compact_period`-th one) moves the survivors.  The stored
model, every counter and the index map must not depend on when the compaction happens: all variants are compared
with the CPU oracle (which compacts at every cull like the reference, src/GlobalModel.cpp:517-579) bit for bit."""
import math
import os

import numpy as np
import pytest

from backends import assert_models_equal, make
from surfelmapping_amd import capi, synth

pytestmark = pytest.mark.gpu

SMALL = dict(width=320, height=120, fx=180.0, fy=180.0, cx=159.5, cy=59.5)
IDENT = np.eye(4, dtype=np.float32).T.reshape(16).copy()
COUNT_KEYS = ("count", "offset", "data_count", "conflict_count", "unstable_count", "fused_count", "visible_count", "tick")


def pair(cam, **over):
    args = (cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    over.setdefault("preprocess", 0)
    return make("oracle", *args, **over), make("hip", *args, **over)


def same_counts(o, h, what):
    co, ch = o.counts(), h.counts()
    assert {k: co[k] for k in COUNT_KEYS} == {k: ch[k] for k in COUNT_KEYS}, what


def wavy(n):
    return [synth.pose_matrix(0.15 * math.sin(0.7 * k), 0.0, 0.35 * k, 0.25 * math.sin(0.5 * k)) for k in range(n)]


@pytest.mark.parametrize("period", [1, 2, 5, 1000])
def test_any_compaction_schedule_gives_the_oracle_model(period):
    seq = synth.make_sequence(SMALL, wavy(36), seed=21, noise_mm=6.0)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=700, compact_period=period)
    for k, fr in enumerate(seq):
        o.process_frame(*fr); h.process_frame(*fr)
        same_counts(o, h, f"period={period} frame {k}")          # counters never include dead slots
    log = h.read_frame_log(64)
    dead_carried = log["n_slots"].astype(np.int64) - log["n_before"].astype(np.int64)
    assert log["n_kill"].sum() > 2000, "the sequence must actually cull"
    if period == 1:
        assert dead_carried.max() == 0                     # compacts at every cull
    else:
        assert dead_carried.max() > 0                      # dead slots were carried over frames ...
    if period in (2, 5):
        again = np.nonzero(dead_carried > 0)[0]
        assert (dead_carried[again[0]:] == 0).any(), dead_carried      # ... and squeezed out by a later cull
    if period == 1000:
        assert dead_carried[-1] > 0                        # still dead slots in the model when it is downloaded
    assert_models_equal(o.download_model(), h.download_model(), f"period={period}")
    same_counts(o, h, "after the download's compaction")
    # and the run continues on the compacted model
    for fr in synth.make_sequence(SMALL, wavy(40)[36:], seed=22, noise_mm=6.0):
        o.process_frame(*fr); h.process_frame(*fr)
    assert_models_equal(o.download_model(), h.download_model(), f"period={period} continued")


def test_index_map_ids_are_positions_among_live_surfels():
    seq = synth.make_sequence(SMALL, wavy(14), seed=23, noise_mm=6.0)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=700, compact_period=1000)
    for fr in seq:
        o.process_frame(*fr); h.process_frame(*fr)
    log = h.read_frame_log(4)
    assert log["n_slots"][-1] > log["n_before"][-1]        # dead slots present: slot != id
    io, ih = o.download_index_map(), h.download_index_map()
    np.testing.assert_array_equal(io[0], ih[0])
    # attributes: the product reads them from the model as it is now, the reference's textures date from the splat,
    # so they agree wherever this frame's fusion did not touch the surfel
    untouched = ih[2][..., 3] != float(h.counts()["tick"] - 1)
    for a, b, name in zip(io[1:], ih[1:], ("vertConf", "colorTime", "normRad")):
        np.testing.assert_array_equal(a.view(np.uint32)[untouched], b.view(np.uint32)[untouched], err_msg=name)
    assert (io[0] > 0).sum() > 1000 and untouched.sum() > 1000
    assert_models_equal(o.download_model(), h.download_model(), "after index-map download")


def test_clean_points_and_reset_with_dead_slots():
    seq = synth.make_sequence(SMALL, wavy(16), seed=24, noise_mm=6.0)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=700, compact_period=1000)
    for fr in seq[:8]:
        o.process_frame(*fr); h.process_frame(*fr)
    rgb, d, s, p = seq[5]
    o.clean_points(d, s, p); h.clean_points(d, s, p)
    co, ch = o.counts(), h.counts()
    assert co["count"] == ch["count"] and co["conflict_count"] == ch["conflict_count"]
    for fr in seq[8:12]:
        o.process_frame(*fr); h.process_frame(*fr)
        same_counts(o, h, "after cleanPoints")
    assert_models_equal(o.download_model(), h.download_model(), "cleanPoints")
    for fr in seq[12:14]:
        o.process_frame(*fr); h.process_frame(*fr)
    o.reset(); h.reset()
    for fr in seq[14:]:
        o.process_frame(*fr); h.process_frame(*fr)
        same_counts(o, h, "after reset")
    a, b = o.download_model(), h.download_model()
    assert a.shape == b.shape
    ok = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))     # raw-cloud normals: NaN payloads differ
    assert ok.all()


def test_capacity_pressure_forces_compaction():
    """Dead slots must never make a frame fail that fits once they are squeezed out (and vice versa)."""
    seq = synth.make_sequence(SMALL, wavy(30), seed=25, noise_mm=6.0)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=260, compact_period=1000)    # 67600 slots, P = 38400
    rcs = []
    for k, fr in enumerate(seq):
        ro = o.process_frame(*fr, allow=(0, -2)); rh = h.process_frame(*fr, allow=(0, -2))
        assert ro == rh, f"frame {k}"
        rcs.append(rh)
        same_counts(o, h, f"frame {k}")
    assert o.counts()["count"] + 38400 > 260 * 260          # the pressure rule was in force
    assert_models_equal(o.download_model(), h.download_model(), "capacity pressure")


def test_conflict_cap_and_id_zero_rule_with_dead_slots():
    """More conflicts than pixels (only the first W*H in surfel order count) while slots are dead,
    and the surfel the reference calls id 0 (never associated, never conflicting) is no longer in slot 0."""
    cam = dict(width=48, height=32, fx=40.0, fy=40.0, cx=23.5, cy=15.5)
    o, h = pair(cam, stereo_border=0.0, conflict_cap=1, max_sqrt_vertices=200, compact_period=1000)
    rng = np.random.default_rng(7)
    n = 20000
    m = synth.seeded_model(n, tick=1, seed=9)
    m[:, 0] = rng.uniform(-1.5, 1.5, n)
    m[:, 1] = rng.uniform(-1.0, 1.0, n)
    m[:, 2] = rng.uniform(3.0, 6.0, n)
    m[:, 3] = rng.uniform(0.5, 3.5, n)
    m[::7, 3] = 0.0                      # dead-already surfels, slot 0 among them
    o.upload_model(m); h.upload_model(m)
    rgb = np.zeros((32, 48, 3), np.uint8)
    sem = np.zeros((32, 48), np.uint8)
    far = np.full((32, 48), 20000, np.uint16)      # everything measured farther -> conflicts
    near = np.full((32, 48), 4500, np.uint16)
    o.process_frame(rgb, far, sem, IDENT); h.process_frame(rgb, far, sem, IDENT)      # reference frame only
    for k, d in enumerate([far, near, far, near, far]):
        o.process_frame(rgb, d, sem, IDENT); h.process_frame(rgb, d, sem, IDENT)
        same_counts(o, h, f"frame {k}")
    log = h.read_frame_log(8)
    assert log["conflict_count"].max() == 48 * 32 and (log["n_slots"] > log["n_before"]).any()
    assert_models_equal(o.download_model(), h.download_model(), "cap + dead slots")


def test_async_frames_with_dead_slots_match_oracle():
    seq = synth.make_sequence(SMALL, wavy(60), seed=26, noise_mm=6.0)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=900)          # default threshold
    P = SMALL["width"] * SMALL["height"]
    bufs = []
    for rgb, d, s, p in seq:
        dr, dd, ds = h.device_alloc(P * 3), h.device_alloc(P * 2), h.device_alloc(P)
        h.device_upload(dr, rgb); h.device_upload(dd, d); h.device_upload(ds, s)
        bufs.append((dr, dd, ds, p))
    for fr in seq:
        o.process_frame(*fr)
    for dr, dd, ds, p in bufs:
        h.process_frame_device(dr, dd, ds, p)
    h.sync()
    same_counts(o, h, "async")
    assert_models_equal(o.download_model(), h.download_model(), "async")


@pytest.mark.parametrize("period", [1000, 3])
def test_conflict_cap_binds_in_asynchronous_frames(period):
    """The one-pass surfel kernel treats every conflict as effective and a fixup kernel takes the ones beyond the first
    W*H (in surfel order) back: resurrect the surfels they killed, restore the confidences they decremented, draw the
    resurrected ones into the index map.  Frames are enqueued without host waits; the cap binds in most of them
    (20 000 surfels in view, 1 536 pixels), confidences are arbitrary floats (exact restore, not +1.0f)."""
    cam = dict(width=48, height=32, fx=40.0, fy=40.0, cx=23.5, cy=15.5)
    o, h = pair(cam, stereo_border=0.0, conflict_cap=1, max_sqrt_vertices=200, compact_period=period)
    rng = np.random.default_rng(11)
    n = 20000
    m = synth.seeded_model(n, tick=1, seed=5)
    m[:, 0] = rng.uniform(-1.5, 1.5, n)
    m[:, 1] = rng.uniform(-1.0, 1.0, n)
    m[:, 2] = rng.uniform(3.0, 6.0, n)
    m[:, 3] = rng.uniform(0.5, 4.5, n).astype(np.float32)
    m[::5, 3] = np.float32(16777218.0)            # conf - 1 is not representable: only an undo plane restores it
    m[3::11, 3] = 0.0
    o.upload_model(m); h.upload_model(m)
    rgb = rng.integers(0, 255, (32, 48, 3), dtype=np.uint8)
    sem = np.zeros((32, 48), np.uint8)
    far = np.full((32, 48), 20000, np.uint16)
    mid = np.full((32, 48), 4500, np.uint16)
    P = 48 * 32
    frames = [far, far, mid, far, far, mid, far, far, far]
    bufs = []
    for d in frames:
        dr, dd, ds = h.device_alloc(P * 3), h.device_alloc(P * 2), h.device_alloc(P)
        h.device_upload(dr, rgb); h.device_upload(dd, d); h.device_upload(ds, sem)
        bufs.append((dr, dd, ds))
    for d in frames:
        o.process_frame(rgb, d, sem, IDENT)
    for b in bufs:
        h.process_frame_device(*b, IDENT)
    h.sync()
    same_counts(o, h, "async, cap binding")
    log = h.read_frame_log(16)
    assert (log["conflict_count"] == P).sum() >= 3, log["conflict_count"]
    if os.environ.get("SM_TWO_LAUNCH") != "0" and os.environ.get("SM_DEFER_ASSOC") != "0":
        # the frames whose cap bound went through the two-launch frame's wait: their association ran in the same launch as the repair
        assert h.debug_slow_frames() >= 3
    assert_models_equal(o.download_model(), h.download_model(), "async, cap binding")
    np.testing.assert_array_equal(o.download_index_map()[0], h.download_index_map()[0])


@pytest.mark.gpu
@pytest.mark.parametrize("cap", [0, 1])
def test_id_zero_dies_in_asynchronous_frames(cap):
    """The surfel the reference addresses as id 0 (conflict.geom:15, data.vert:142) is dead on arrival (conf 0, an uploaded
    model) and so are its successors: the pass sees it die, the publisher has to find the next live slot BEFORE the frame's
    association may use the exemption -- in the two-launch frame both run in the same launch and the association waits.  Frames
    are enqueued without host waits, with the W*H cap off and (binding) on."""
    cam = dict(width=48, height=32, fx=40.0, fy=40.0, cx=23.5, cy=15.5)
    o, h = pair(cam, stereo_border=0.0, conflict_cap=cap, max_sqrt_vertices=200, compact_period=1000, fuse_thresh=0.05)
    rng = np.random.default_rng(23)
    n = 6000
    m = synth.seeded_model(n, tick=1, seed=4)
    m[:, 0] = rng.uniform(-1.5, 1.5, n)
    m[:, 1] = rng.uniform(-1.0, 1.0, n)
    m[:, 2] = rng.uniform(3.0, 6.0, n)
    m[:, 3] = rng.uniform(0.5, 3.5, n)
    m[:40, 3] = 0.0                      # id 0 and the 39 slots behind it are dead already
    m[40, 3] = 0.7                       # the first live one dies at its first conflict: the exemption moves twice
    o.upload_model(m); h.upload_model(m)
    rgb = rng.integers(0, 255, (32, 48, 3), dtype=np.uint8)
    sem = np.zeros((32, 48), np.uint8)
    far = np.full((32, 48), 20000, np.uint16)
    mid = np.full((32, 48), 4500, np.uint16)
    P = 48 * 32
    frames = [far, mid, far, far, mid, far]
    bufs = []
    for d in frames:
        dr, dd, ds = h.device_alloc(P * 3), h.device_alloc(P * 2), h.device_alloc(P)
        h.device_upload(dr, rgb); h.device_upload(dd, d); h.device_upload(ds, sem)
        bufs.append((dr, dd, ds))
    for d in frames:
        o.process_frame(rgb, d, sem, IDENT)
    for b in bufs:
        h.process_frame_device(*b, IDENT)
    h.sync()
    same_counts(o, h, f"async, id 0 dies, cap={cap}")
    if os.environ.get("SM_TWO_LAUNCH") != "0" and os.environ.get("SM_DEFER_ASSOC") != "0":
        assert h.debug_slow_frames() >= 1
    assert_models_equal(o.download_model(), h.download_model(), f"async, id 0 dies, cap={cap}")
    np.testing.assert_array_equal(o.download_index_map()[0], h.download_index_map()[0])


@pytest.mark.parametrize("period,thresh", [(3, 0.05), (8, 0.0), (1000, 0.05)])
def test_async_frames_hand_their_association_to_the_next_frame(period, thresh):
    """Asynchronous plain streams hold a frame's association back and launch it together with the next frame's image
    preparation (k_assoc_prep); compacting frames, synchronisations and downloads in between take it out first.  Counters
    of every frame (device-side frame log), a model in the middle and the model at the end against the oracle."""
    seq = synth.make_sequence(SMALL, wavy(40), seed=31, noise_mm=3.0)
    o, h = pair(SMALL, stereo_border=20.0, max_sqrt_vertices=900, fuse_thresh=thresh, compact_period=period)
    P = SMALL["width"] * SMALL["height"]
    bufs = []
    for rgb, d, s_, p in seq:
        dr, dd, ds = h.device_alloc(P * 3), h.device_alloc(P * 2), h.device_alloc(P)
        h.device_upload(dr, rgb); h.device_upload(dd, d); h.device_upload(ds, s_)
        bufs.append((dr, dd, ds, p))
    ref = []
    for k, fr in enumerate(seq):
        o.process_frame(*fr)
        ref.append(o.counts())
        if k == 17:
            mid = o.download_model()
    for k, b in enumerate(bufs):
        h.process_frame_device(*b)
        if k == 9:
            h.sync()                                   # a wait in the middle: the held-back association runs alone
            assert all(h.counts()[x] == ref[9][x] for x in ("count", "offset", "unstable_count", "fused_count", "conflict_count"))
        if k == 17:
            assert_models_equal(mid, h.download_model(), "in the middle")
    h.sync()
    same_counts(o, h, "end")
    log = h.read_frame_log(64)
    assert len(log) == len(seq) - 1                    # every fusing frame
    for e, r in zip(log, ref[1:]):
        assert (e["unstable_count"], e["fused_count"], e["conflict_count"]) == (r["unstable_count"], r["fused_count"], r["conflict_count"])
    assert_models_equal(o.download_model(), h.download_model(), "end")
    if thresh > 0:
        assert sum(r["fused_count"] for r in ref) > 1000


def fixup_only_tile_scenario():
    """A model and frames where a tile is drawn ONLY through k_pass_fixup: 16 x 8 pixels (cap = 128 conflicts),
    every pixel class 10 (sky: conflict.vert:51-54 measures max + 1 for it, so every surfel in view conflicts) at 4 m.
    Tile 0 (slots 0..1023): 1024 class-0 surfels in view at 5 m -- the first 128 conflicts in slot order use the cap up.
    Tile 1: three class-10 surfels exactly on the rays of checkerboard pixels at 4 m, conf 0.9, last updated at time 1: the
    pass kills them (conflict, 0.9 - 1 <= 0), the cap takes that back (resurrected, drawn by the fixup), and the association
    of the same frame fuses them (same depth, same class, data.vert:151) -- which sets their time to 3.  time_delta = 2: from
    frame 4 on, the tile's stale time word (1) would gate it out of the index map while its surfels (3) still belong there."""
    W, H = 16, 8
    fx, cxx, cyy = 100.0, W / 2 - 0.5, H / 2 - 0.5

    def on_ray(i, j, z, **kw):
        s = np.zeros(12, np.float32)
        s[0] = np.float32(np.float32(np.float32(i + 0.5) - np.float32(cxx)) * np.float32(z)) * np.float32(1.0 / fx)
        s[1] = np.float32(np.float32(np.float32(j + 0.5) - np.float32(cyy)) * np.float32(z)) * np.float32(1.0 / fx)
        s[2] = z
        s[3] = kw.get("conf", 0.9)
        s[4] = np.array([np.uint32(kw.get("sem", 0) << 24 | 0x102030)], np.uint32).view(np.float32)[0]
        s[6], s[7] = 1.0, 1.0
        s[10] = 1.0
        s[11] = 0.05
        return s

    tile0 = [on_ray(k % W, (k // W) % H, 5.0) for k in range(1024)]
    tile1 = [on_ray(i, j, 4.0, sem=10) for i, j in ((3, 2), (8, 5), (11, 4))]          # (i + j) odd: data.vert:88
    model = np.stack(tile0 + tile1)
    rgb = np.zeros((H, W, 3), np.uint8)
    rgb[..., 0], rgb[..., 1], rgb[..., 2] = 0x10, 0x20, 0x30
    sky4 = (rgb, np.full((H, W), 4000, np.uint16), np.full((H, W), 10, np.uint8))
    road4 = (rgb, np.full((H, W), 4000, np.uint16), np.zeros((H, W), np.uint8))        # class 0 at the same depth: nothing conflicts
    return dict(width=W, height=H, fx=fx, fy=fx, cx=cxx, cy=cyy), model, [sky4, road4, road4, road4]


@pytest.mark.parametrize("asynchronous", [False, True])
def test_tile_drawn_only_through_the_cap_fixup_keeps_its_time_stamp(asynchronous):
    cam, model, frames = fixup_only_tile_scenario()
    o, h = pair(cam, stereo_border=0.0, conflict_cap=1, max_sqrt_vertices=64, compact_period=1000, time_delta=2)
    P = cam["width"] * cam["height"]
    for b in (o, h):
        b.upload_model(model)
        b.set_tick(3)
    ref = []
    for fr in frames:
        o.process_frame(*fr, IDENT)
        ref.append((o.counts(), o.download_index_map()[0].copy()))
    assert ref[0][0]["conflict_count"] == P and ref[0][0]["fused_count"] == 3, ref[0][0]       # the scenario does what it says
    ids1 = set(range(1024 - 128, 1024 - 128 + 3))          # tile 1's surfels after frame 3's cull (128 of tile 0 died)
    assert ids1 <= set(ref[1][1].ravel().tolist()), "frame 4 must still draw tile 1's surfels (updated at 3, delta 2)"
    if asynchronous:
        bufs = []
        for rgb, d, s in frames:
            dr, dd, ds = h.device_alloc(P * 3), h.device_alloc(P * 2), h.device_alloc(P)
            h.device_upload(dr, rgb); h.device_upload(dd, d); h.device_upload(ds, s)
            bufs.append((dr, dd, ds))
        for b in bufs:
            h.process_frame_device(*b, IDENT)
        h.sync()
        log = h.read_frame_log(8)
        assert [int(x) for x in log["visible_count"]] == [r[0]["visible_count"] for r in ref], log["visible_count"]
        assert [int(x) for x in log["fused_count"]] == [r[0]["fused_count"] for r in ref]
    else:
        for k, fr in enumerate(frames):
            h.process_frame(*fr, IDENT)
            ch = h.counts()
            assert {x: ch[x] for x in COUNT_KEYS} == {x: ref[k][0][x] for x in COUNT_KEYS}, f"frame {k}"
            np.testing.assert_array_equal(h.download_index_map()[0], ref[k][1], err_msg=f"index map, frame {k}")
    same_counts(o, h, "end")
    assert_models_equal(o.download_model(), h.download_model(), "fixup-only tile")
