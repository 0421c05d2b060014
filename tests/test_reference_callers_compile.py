"""The reference's own callers parse and type-check against the drop-in facade, unchanged.

`build_map.cpp` and `load_map.cpp` (and, through them, the reference's `gui/GUI.h` and `src/Utils/Checker.h`) are read IN
PLACE from /root/reference -- nothing of the reference is copied into this repository -- and compiled with
`g++ -fsyntax-only` against `surfelmapping_amd/csrc/facade/` (SurfelMapping, GlobalModel, IndexMap, FeedbackBuffer,
Config, GPUTexture, KittiReader) plus the declaration-only Eigen / Pangolin / OpenCV stand-ins of tests/stubs/ (those
libraries are not installed here; see tests/stubs/README.md).  Zero edits to the callers: every method they call --
processFrame, cleanPoints, reset, acquireImages, getTexture, getFeedbackBuffer(RAW)->render, getGlobalModel().renderModel /
getModelMapNR / downloadMap / uploadMap, getCurrPose, getHistoryPoses, Config::*, KittiReader::* -- must exist with a
compatible signature.  Skipped where the reference mount does not exist (the GPU box)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference mount absent")
@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("gl", [False, True])
@pytest.mark.parametrize("unit", ["build_map.cpp", "load_map.cpp"])
def test_reference_caller_compiles_against_the_facade(unit, gl):
    """gl = True: with -DSM_FACADE_GL, i.e. the branches that fill real GL objects (vertex buffers for renderModel and the raw
    cloud, pangolin::GlTexture::Reinitialise / Upload for getTexture and the mirror planes) are type-checked too, against the GL /
    Pangolin declarations of tests/stubs/pangolin_stub.h."""
    src = os.path.join(REF, unit)
    assert os.path.exists(src)
    cmd = ["g++", "-std=c++17", "-fsyntax-only"] + (["-DSM_FACADE_GL"] if gl else []) + [ "-Wall", "-Wno-format", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
           "-Wno-sign-compare", "-Wno-unused-parameter",
           "-I", os.path.join(ROOT, "surfelmapping_amd", "csrc", "facade"),     # SurfelMapping.h, KittiReader.h, Config.h ... (first: drop-in)
           "-I", os.path.join(ROOT, "tests", "stubs"),                           # Eigen / pangolin / opencv2 declarations, Shaders.h
           "-I", os.path.join(REF, "gui"), "-I", os.path.join(REF, "src", "Utils"),   # the reference's GUI.h and Checker.h, in place
           src]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "error" not in r.stderr
