"""The drop-in KittiReader (surfelmapping_amd/csrc/facade/KittiReader.h) on a synthetic dataset in the reference's
directory layout: decoded frames equal the arrays that were written (CPU), and build_map's loop on top of it produces
the oracle's map (GPU)."""
import os
import subprocess

import numpy as np
import pytest

import kitti_fixture as kf
from surfelmapping_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAM = dict(width=320, height=120, fx=180.0, fy=180.0, cx=159.5, cy=59.5)


def fnv(b: bytes) -> int:
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def build(tmp_path):
    exe = str(tmp_path / "kitti_demo")
    lib = os.path.join(ROOT, "surfelmapping_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "kitti_demo.cpp"),
                           "-L" + lib, "-lsurfelmapping_hip", "-lz", "-Wl,-rpath," + lib])
    return exe


def dataset(tmp_path, n=4, seed=17):
    poses = synth.kitti_trajectory(n)
    seq = synth.make_sequence(CAM, poses, seed=seed, noise_mm=2.0)
    root = str(tmp_path / "kitti")
    kf.write_dataset(root, CAM, seq, poses)
    return root, seq, poses


def test_reader_decodes_every_filter_type_and_the_text_files(tmp_path):
    root, seq, poses = dataset(tmp_path, n=3)
    r = subprocess.run([build(tmp_path), root], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "calib 180 180 159.5 59.5 320 120 frames 3"
    for k, (rgb, depth, sem, _) in enumerate(seq):
        want = (f"frame {k} t={0.1 * k:.3f} rgb {fnv(rgb.tobytes()):016x} depth {fnv(depth.tobytes()):016x} "
                f"sem {fnv(sem.tobytes()):016x} pose {fnv(kf.reader_pose(poses[k]).tobytes()):016x}")
        assert lines[1 + k] == want


@pytest.mark.gpu
def test_build_map_loop_on_kitti_layout_matches_oracle(tmp_path):
    import oracle_lib as ol
    root, seq, poses = dataset(tmp_path, n=5)
    out = tmp_path / "map.bin"
    r = subprocess.run([build(tmp_path), root, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    o = ol.Oracle(ol.make_config(**CAM, preprocess=1, max_sqrt_vertices=1000))
    for k, (rgb, depth, sem, _) in enumerate(seq):
        o.process_frame(rgb, depth, sem, kf.reader_pose(poses[k]))
    raw = open(out, "rb").read()
    n = int(np.frombuffer(raw[:4], np.uint32)[0])
    ref = o.download_model()
    assert n == ref.shape[0] > 1000
    assert np.array_equal(np.frombuffer(raw[12:], np.uint32).reshape(n, 12), ref.view(np.uint32))
