#!/usr/bin/env python3
"""Regenerates tests/golden/*.json: regression pins of THIS repo's oracle on seeded synthetic sequences.

They are NOT reference outputs -- the reference's GL path cannot be built or run here and ships no fixtures
("parity unpinned"); they only make silent drift of the oracle (and, through the parity tests, of the
HIP path) visible.  Inputs are regenerated from the seeds by surfelmapping_amd/synth.py; each file stores the
per-frame counters and the SHA-256 of the model's raw bytes after the last frame."""
import hashlib
import json
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol                      # noqa: E402
from surfelmapping_amd import synth         # noqa: E402

CASES = {
    "moving_small": dict(cam=dict(width=320, height=120, fx=180.0, fy=180.0, cx=159.5, cy=59.5), n=8, seed=3, noise=0.0,
                         over=dict(preprocess=0, stereo_border=20.0), traj="kitti"),
    "fuse_thresh_noise": dict(cam=dict(width=320, height=120, fx=180.0, fy=180.0, cx=159.5, cy=59.5), n=8, seed=5, noise=4.0,
                              over=dict(preprocess=0, stereo_border=20.0, fuse_thresh=0.05), traj="slow"),
    "preprocess_chain": dict(cam=dict(width=320, height=120, fx=180.0, fy=180.0, cx=159.5, cy=59.5), n=6, seed=31, noise=3.0,
                             over=dict(preprocess=1, stereo_border=20.0), traj="kitti"),
}


def sequence(case):
    if case["traj"] == "kitti":
        poses = synth.kitti_trajectory(case["n"])
    else:
        poses = [synth.pose_matrix(0, 0, 0.05 * k, 0.2 * math.sin(k)) for k in range(case["n"])]
    return synth.make_sequence(case["cam"], poses, seed=case["seed"], noise_mm=case["noise"])


def run(case, make):
    m = make(case)
    hist = []
    for fr in sequence(case):
        m.process_frame(*fr)
        c = m.counts()
        hist.append([c[k] for k in ("count", "offset", "data_count", "conflict_count", "unstable_count", "fused_count")])
    return hist, hashlib.sha256(m.download_model().tobytes()).hexdigest()


def main():
    for name, case in CASES.items():
        hist, digest = run(case, lambda c: ol.Oracle(ol.make_config(**c["cam"], **c["over"])))
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump({"case": case, "counts_per_frame": hist, "model_sha256": digest}, f, indent=1)
        print(name, hist[-1], digest[:16])


if __name__ == "__main__":
    main()
