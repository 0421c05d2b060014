"""The C++ drop-in facade (surfelmapping_amd/csrc/facade/*.h) used the way build_map.cpp uses the
reference classes.  CPU: it compiles with plain g++ against the C-ABI only.  GPU: the binary's
saved map equals the oracle's model bit for bit."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_demo.cpp")
LIBDIR = os.path.join(ROOT, "surfelmapping_amd")


def build_demo(tmp_path):
    exe = str(tmp_path / "facade_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", exe, SRC,
                           "-L" + LIBDIR, "-lsurfelmapping_hip", "-Wl,-rpath," + LIBDIR])
    return exe


def test_facade_compiles_against_c_abi_only(tmp_path):
    exe = build_demo(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stdout


@pytest.mark.gpu
def test_facade_demo_matches_oracle(tmp_path):
    import oracle_lib as ol
    from surfelmapping_amd import synth
    cam = dict(width=320, height=120, fx=180.0, fy=180.0, cx=159.5, cy=59.5)
    seq = synth.make_sequence(cam, synth.kitti_trajectory(5), seed=12)
    frames = tmp_path / "frames.bin"
    with open(frames, "wb") as f:
        f.write(np.array([cam["width"], cam["height"], len(seq)], np.uint32).tobytes())
        f.write(np.array([cam["fx"], cam["fy"], cam["cx"], cam["cy"]], np.float32).tobytes())
        for rgb, d, s, p in seq:
            f.write(rgb.tobytes()); f.write(d.tobytes()); f.write(s.tobytes()); f.write(p.astype(np.float32).tobytes())
    exe = build_demo(tmp_path)
    out = tmp_path / "map.bin"
    r = subprocess.run([exe, str(frames), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    o = ol.Oracle(ol.make_config(**cam, preprocess=0, max_sqrt_vertices=1000))
    for fr in seq:
        o.process_frame(*fr)
    raw = open(out, "rb").read()
    n = int(np.frombuffer(raw[:4], np.uint32)[0])
    m = np.frombuffer(raw[12:], np.float32).reshape(n, 12)
    ref = o.download_model()
    assert n == ref.shape[0] > 0
    assert np.array_equal(m.view(np.uint32), ref.view(np.uint32))
    c = o.counts()
    assert f"frame 4: model {c['count']} offset {c['offset']} data {c['data_count']} conflict {c['conflict_count']}" in r.stdout
