"""Differential fuzzing of the HIP path against the CPU oracle: seeded random scenarios mixing every frame-level entry
point (processFrame, cleanPoints, reset, map upload / download, index-map download) on small images, with hostile
inputs (NaN / infinite / negative confidences and positions in uploaded maps, zero and out-of-range depth, every
compaction period).  After every step the counters agree; at every download the stored model is bit-identical."""
import math
import os

import numpy as np
import pytest

from backends import assert_models_equal, make
from surfelmapping_amd import synth

pytestmark = pytest.mark.gpu

COUNT_KEYS = ("count", "offset", "data_count", "conflict_count", "unstable_count", "fused_count", "visible_count", "tick")


def hostile_model(rng, n, tick):
    m = synth.seeded_model(n, tick=tick, seed=int(rng.integers(1 << 30)))
    m[:, 0] = rng.uniform(-2.0, 2.0, n)
    m[:, 1] = rng.uniform(-1.5, 1.5, n)
    m[:, 2] = rng.uniform(0.5, 9.0, n)
    m[:, 3] = rng.choice(np.array([0.9, 1.8, 2.7, 0.0, -1.0, 1.0, 0.1], np.float32), n)
    bad = rng.random(n) < 0.02
    m[bad, 3] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), int(bad.sum()))
    bad = rng.random(n) < 0.01
    m[bad, int(rng.integers(0, 3))] = rng.choice(np.array([np.nan, np.inf, -np.inf, 1e30], np.float32), int(bad.sum()))
    return m


def random_frame(rng, W, H):
    base = rng.uniform(2.0, 7.0)
    yy, xx = np.mgrid[0:H, 0:W]
    d = base + 0.8 * np.sin(xx / rng.uniform(5, 15)) + 0.5 * np.cos(yy / rng.uniform(4, 12)) + rng.normal(0, 0.01, (H, W))
    d[rng.random((H, W)) < 0.03] = 0.0                       # holes
    if rng.random() < 0.3:
        d[: H // 3] = 0.0                                    # sky
    if rng.random() < 0.2:
        d[:, -W // 4:] = 40.0                                # beyond far clip (metricise drops it)
    depth = np.clip(d * 1000.0, 0, 65535).astype(np.uint16)
    sem = rng.integers(0, 19, (H, W), dtype=np.uint8)
    if rng.random() < 0.5:
        sem[:] = np.uint8(rng.integers(0, 19))               # large uniform regions let the filters / fusion bite
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    return rgb, depth, sem


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("SM_FUZZ_SEEDS", "64")))))   # SM_FUZZ_SEEDS=2000 for a soak
def test_random_call_sequences_match_the_oracle(seed, tmp_path):
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.choice([48, 64, 96])), int(rng.choice([32, 48]))
    cam = dict(width=W, height=H, fx=0.8 * W, fy=0.8 * W, cx=W / 2 - 0.5, cy=H / 2 - 0.5)
    over = dict(preprocess=int(rng.integers(0, 2)), stereo_border=float(rng.choice([0.0, 6.0])),
                conflict_cap=int(rng.integers(0, 2)), fuse_thresh=float(rng.choice([0.0, 0.02, 0.2])),
                max_sqrt_vertices=int(rng.choice([120, 200, 400])), time_delta=int(rng.choice([3, 200])))
    args = (W, H, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    o = make("oracle", *args, **over)
    h = make("hip", *args, compact_period=int(rng.choice([1, 2, 3, 8, 1000])), **over)
    cap = over["max_sqrt_vertices"] ** 2
    z = 0.0
    last = None
    for step in range(60):
        what = rng.choice(["frame"] * 10 + ["burst"] * 2 + ["clean", "reset", "upload", "download", "index", "render", "saveload"])
        tag = f"seed {seed} step {step} {what}"
        if what == "frame":
            z += rng.uniform(-0.05, 0.25)
            pose = synth.pose_to_colmajor(synth.pose_matrix(0.05 * math.sin(step), 0.0, z, float(rng.uniform(-3, 3))))
            fr = random_frame(rng, W, H)
            ro = o.process_frame(*fr, pose, allow=(0, -2))
            rh = h.process_frame(*fr, pose, allow=(0, -2))
            assert ro == rh, tag
            last = (fr, pose)
        elif what == "burst":
            # several frames enqueued without waiting (device-resident inputs), then one sync
            P = W * H
            for _ in range(int(rng.integers(2, 6))):
                z += rng.uniform(-0.05, 0.25)
                pose = synth.pose_to_colmajor(synth.pose_matrix(0.05 * math.sin(step), 0.0, z, float(rng.uniform(-3, 3))))
                fr = random_frame(rng, W, H)
                ro = o.process_frame(*fr, pose, allow=(0, -2))
                dr, dd, ds = h.device_alloc(P * 3), h.device_alloc(P * 2), h.device_alloc(P)
                h.device_upload(dr, fr[0]); h.device_upload(dd, fr[1]); h.device_upload(ds, fr[2])
                h.process_frame_device(dr, dd, ds, pose)
                last = (fr, pose)
                if ro != 0:
                    break                                   # capacity error on the oracle: let the sync below report it too
            rh = h.sync(allow=(0, -2))
            assert (ro != 0) == (rh != 0), tag
        elif what == "render" and o.counts()["count"] > 0:
            view = synth.pose_to_colmajor(synth.pose_matrix(0.1, 0.0, z - 0.5, 2.0))
            a = o.render_image(view, 40, 30, 30.0, 30.0, 19.5, 14.5)
            b = h.render_image(view, 40, 30, 30.0, 30.0, 19.5, 14.5)
            np.testing.assert_array_equal(a[0], b[0], err_msg=tag); np.testing.assert_array_equal(a[1], b[1], err_msg=tag)
        elif what == "saveload":
            path = str(tmp_path / f"m{step}.bin")
            h.save_map(path, 1, 2)
            raw = open(path, "rb").read()
            n = int(np.frombuffer(raw[:4], np.uint32)[0])
            assert n == o.counts()["count"], tag
            assert_models_equal(np.frombuffer(raw[12:], np.float32).reshape(n, 12), o.download_model(), tag)
            assert h.load_map(path) == (1, 2)
            o.upload_model(np.frombuffer(raw[12:], np.float32).reshape(n, 12).copy())
        elif what == "clean" and last is not None:
            fr, pose = last
            o.clean_points(fr[1], fr[2], pose); h.clean_points(fr[1], fr[2], pose)
        elif what == "reset":
            o.reset(); h.reset()
        elif what == "upload":
            n = int(rng.integers(0, min(cap // 2, 30000)))
            m = hostile_model(rng, n, tick=max(o.counts()["tick"], 1))
            o.upload_model(m); h.upload_model(m)
        elif what == "download":
            assert_models_equal(o.download_model(), h.download_model(), tag)
        elif what == "index":
            np.testing.assert_array_equal(o.download_index_map()[0], h.download_index_map()[0], err_msg=tag)
        co, ch = o.counts(), h.counts()
        if what in ("frame", "burst", "upload", "download", "saveload"):
            assert {k: co[k] for k in COUNT_KEYS} == {k: ch[k] for k in COUNT_KEYS}, tag
        else:
            assert co["count"] == ch["count"] and co["tick"] == ch["tick"], tag
    a, b = o.download_model(), h.download_model()
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))       # reset re-init: NaN payloads of raw normals
    assert a.shape == b.shape and same.all(), f"seed {seed} final"
