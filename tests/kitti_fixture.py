"""Writes a tiny dataset in the reference's KITTI layout (gui/KittiReader.cpp:20-44): image_2/, PSMNet/, semantics/,
calibration.txt, pose.txt, times.txt.  PNGs are produced here with every filter type so that the reader's decoder is
exercised (the reference relies on cv::imread)."""
import os
import struct
import zlib

import numpy as np


def _filter_row(ft, row, prev, bpp):
    row = row.astype(np.int32)
    prev = prev.astype(np.int32)
    a = np.concatenate([np.zeros(bpp, np.int32), row[:-bpp]])
    c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
    if ft == 0:
        pred = 0
    elif ft == 1:
        pred = a
    elif ft == 2:
        pred = prev
    elif ft == 3:
        pred = (a + prev) >> 1
    else:
        p = a + prev - c
        pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
        pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
    return ((row - pred) & 0xFF).astype(np.uint8)


def write_png(path, arr):
    """arr: uint8 [h][w] / [h][w][3] or uint16 [h][w]; rows cycle through filter types 0..4."""
    arr = np.ascontiguousarray(arr)
    h, w = arr.shape[:2]
    if arr.dtype == np.uint16:
        raw, ctype, depth, bpp = arr.astype(">u2").view(np.uint8).reshape(h, w * 2), 0, 16, 2
    elif arr.ndim == 3:
        raw, ctype, depth, bpp = arr.reshape(h, w * 3), 2, 8, 3
    else:
        raw, ctype, depth, bpp = arr.reshape(h, w), 0, 8, 1
    rows = []
    prev = np.zeros(raw.shape[1], np.uint8)
    for y in range(h):
        ft = y % 5
        rows.append(bytes([ft]) + _filter_row(ft, raw[y], prev, bpp).tobytes())
        prev = raw[y]

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    idat = zlib.compress(b"".join(rows), 6)
    half = len(idat) // 2                      # two IDAT chunks: the decoder must concatenate them
    data = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
            chunk(b"IDAT", idat[:half]) + chunk(b"IDAT", idat[half:]) + chunk(b"IEND", b""))
    with open(path, "wb") as f:
        f.write(data)


def write_dataset(root, cam, seq, poses):
    """seq: [(rgb, depth_u16, sem, _)], poses: 4x4 camera->world matrices BEFORE the reader's T20 offset."""
    for d in ("image_2", "PSMNet", "semantics"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    with open(os.path.join(root, "calibration.txt"), "w") as f:
        f.write(f"{cam['fx']} {cam['fy']} {cam['cx']} {cam['cy']}\n{cam['width']} {cam['height']}\n")
    with open(os.path.join(root, "times.txt"), "w") as f:
        for k in range(len(seq)):
            f.write(f"{0.1 * k:.6f}\n")
    with open(os.path.join(root, "pose.txt"), "w") as f:
        for p in poses:
            f.write(" ".join(f"{v:.9g}" for v in np.asarray(p, np.float32)[:3, :].reshape(-1)) + "\n")
    for k, (rgb, depth, sem, _) in enumerate(seq):
        write_png(os.path.join(root, "image_2", f"{k:06d}.png"), rgb)
        write_png(os.path.join(root, "PSMNet", f"{k:06d}.png"), depth)
        write_png(os.path.join(root, "semantics", f"{k:06d}.png"), sem)


def reader_pose(p):
    """What KittiReader hands to processFrame: p * T20 (x offset -0.06 m), evaluated in fp32 as the facade does."""
    g = np.asarray(p, np.float32).copy()
    out = g.copy()
    out[:3, 3] = g[:3, 0] * np.float32(-0.06) + g[:3, 3]
    return np.ascontiguousarray(out.T.reshape(16))
