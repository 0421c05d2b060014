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
@pytest.mark.parametrize("facade_async", ["0", "1"])      # SM_FACADE_ASYNC=1: processFrame only enqueues, the getters wait
def test_facade_demo_matches_oracle(tmp_path, facade_async):
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
    views = tmp_path / "views"
    views.mkdir()
    tex = tmp_path / "textures.bin"
    r = subprocess.run([exe, str(frames), str(out), str(views), str(tex)], capture_output=True, text=True,
                       env=dict(os.environ, SM_FACADE_ASYNC=facade_async))
    assert r.returncode == 0, r.stdout + r.stderr
    o = ol.Oracle(ol.make_config(**cam, preprocess=0, max_sqrt_vertices=1000))
    for fr in seq:
        o.process_frame(*fr)
    # getTexture(): the three float images read back from the core, the three input images as the last processFrame left them
    P = cam["width"] * cam["height"]
    tb = open(tex, "rb").read()
    assert len(tb) == P * (3 * 4 + 3 + 1 + 2)
    for k, which in enumerate((0, 1, 2)):                           # DEPTH_METRIC, DEPTH_FILTERED, LAST
        got = np.frombuffer(tb[k * P * 4:(k + 1) * P * 4], np.float32).reshape(cam["height"], cam["width"])
        want = o.download_depth(which)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), which
    off = 3 * P * 4
    assert np.array_equal(np.frombuffer(tb[off:off + 3 * P], np.uint8), seq[-1][0].ravel())
    assert np.array_equal(np.frombuffer(tb[off + 3 * P:off + 4 * P], np.uint8), seq[-1][2].ravel())
    assert np.array_equal(np.frombuffer(tb[off + 4 * P:], np.uint16), seq[-1][1].ravel())
    raw = open(out, "rb").read()
    n = int(np.frombuffer(raw[:4], np.uint32)[0])
    m = np.frombuffer(raw[12:], np.float32).reshape(n, 12)
    ref = o.download_model()
    assert n == ref.shape[0] > 0
    assert np.array_equal(m.view(np.uint32), ref.view(np.uint32))
    # acquireImages wrote image/000007.png + semantic/000007.png: decode the stored-deflate PNGs and compare
    import struct
    import zlib

    def read_png(path):
        b = open(path, "rb").read()
        assert b[:8] == b"\x89PNG\r\n\x1a\n"
        pos, idat, hdr = 8, b"", None
        while pos < len(b):
            n, typ = struct.unpack(">I4s", b[pos:pos + 8])
            body = b[pos + 8:pos + 8 + n]
            assert zlib.crc32(typ + body) == struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0]
            if typ == b"IHDR":
                hdr = struct.unpack(">IIBBBBB", body)
            if typ == b"IDAT":
                idat += body
            pos += 12 + n
        w, h, depth, ctype = hdr[:4]
        ch = 3 if ctype == 2 else 1
        raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * ch + 1)
        assert depth == 8 and np.all(raw[:, 0] == 0)
        return raw[:, 1:].reshape(h, w, ch)

    bgr, sem = o.render_image(seq[-1][3], cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    img = read_png(str(views / "image" / "000007.png"))
    lab = read_png(str(views / "semantic" / "000007.png"))
    assert np.array_equal(img, bgr[..., ::-1]) and np.array_equal(lab[..., 0], sem) and (sem > 0).mean() > 0.3
    c = o.counts()
    # FeedbackBuffer RAW -> render, GlobalModel::renderModel, getModelMapNR (build_map.cpp:177-204)
    assert f"raw cloud {o.download_raw_cloud().shape[0]}  drawn {c['count']}  mirror 1000x1000  textures 320 120" in r.stdout, r.stdout[-600:]
    assert f"frame 4: model {c['count']} offset {c['offset']} data {c['data_count']} conflict {c['conflict_count']}" in r.stdout
