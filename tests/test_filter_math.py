"""CPU checks of the two conservative filters the HIP path puts in front of exact tests (numpy fp32 restatements of the
device expressions; the kernels themselves are checked bit-exact against the oracle by the -m gpu tests):

* the key-depth filter of the association (sm_kernels.h associate_pixel): the 24-bit depth in the key bounds the winner's
  camera-frame z, so a pixel further from it than the threshold + slack can skip the gather -- it must never skip a pixel
  that data.vert:151's depth test would pass;
* the side-plane test on tile boxes that straddle the camera plane (plane_guard / tile_flags_*): a box is culled only if no
  point of it can pass the exact view tests.
"""
import numpy as np

F = np.float32


def d24_of(z, cutoff):
    """index_map.vert / splat_one: zn = z / cutoff; zw = 0.5 zn + 0.5; d24 = floor(double(zw) * 16777215 + 0.5)"""
    zn = (z / cutoff).astype(F)
    zw = (F(0.5) * zn + F(0.5)).astype(F)
    return np.floor(zw.astype(np.float64) * 16777215.0 + 0.5).astype(np.uint32)


def test_key_depth_filter_never_rejects_a_pixel_the_exact_test_passes():
    rng = np.random.default_rng(5)
    n = 2_000_000
    for cutoff in (F(5.0), F(30.0), F(200.0)):
        z_s = rng.uniform(0.01, float(cutoff) * 0.999, n).astype(F)               # the surfel's camera-frame z (drawn: zn <= 1)
        lam = rng.uniform(1.0, 1.6, n).astype(F)
        for thr in (F(0.0), F(0.003), F(0.05), F(0.5)):
            # measured depths: mostly close to the surfel (the interesting zone), some far
            dz = np.where(rng.random(n) < 0.7, rng.normal(0.0, float(thr) + 2e-3, n), rng.normal(0.0, 1.0, n)).astype(F)
            dz[: n // 10] = 0.0                                                   # exact hits (the static-camera case)
            z_m = (z_s + dz).astype(F)
            exact = np.abs((z_s * lam).astype(F) - (z_m * lam).astype(F)) <= thr            # data.vert:151 (depth part)
            d24 = d24_of(z_s, cutoff)
            z_key = ((d24.astype(F) * F(2.0 / 16777215.0)).astype(F) - F(1.0)).astype(F) * cutoff
            slack = ((F(1.0e-3) + F(1.0e-5) * cutoff) * lam).astype(F)
            rejected = (np.abs((z_key - z_m).astype(F)) * lam).astype(F) > (thr + slack).astype(F)
            assert not np.any(exact & rejected), (float(cutoff), float(thr))
            if thr <= F(0.003):                                                   # ... and it does filter
                assert rejected.mean() > 0.2


def _plane_guard(fx, fy, cols, rows, t, lo, hi):
    S = (np.abs(lo[:, 0]) + np.abs(hi[:, 0])) + (np.abs(lo[:, 1]) + np.abs(hi[:, 1])) + (np.abs(lo[:, 2]) + np.abs(hi[:, 2])) + \
        (abs(t[0]) + abs(t[1]) + abs(t[2]))
    return (F(2.0e-6) * (((fx + fy) + cols) + rows) * S).astype(F)


def test_side_planes_on_straddling_boxes_never_cull_a_visible_point():
    rng = np.random.default_rng(9)
    fx, fy, cx, cy, cols, rows = F(718.856), F(718.856), F(607.19), F(185.2), F(1242.0), F(375.0)
    border = F(80.0)
    culled_any = 0
    for trial in range(40):
        yaw = rng.uniform(-np.pi, np.pi)
        R = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]], dtype=np.float64)
        tw = rng.uniform(-300, 300, 3)
        Ri = R.T.astype(F)
        ti = (-(R.T @ tw)).astype(F)                                              # world -> camera
        nb = 4000
        c = (tw + rng.uniform(-15, 15, (nb, 3)) * np.array([1, 0.3, 1])).astype(F)
        hs = np.abs(rng.normal(0, 2.0, (nb, 3))).astype(F) + F(0.01)
        tiny = rng.random(nb) < 0.3                                               # tiny boxes near the camera plane
        hs[tiny] = rng.uniform(0.002, 0.05, (tiny.sum(), 3)).astype(F)
        c[tiny] = (tw + (R @ (rng.uniform(-0.2, 0.2, (tiny.sum(), 3)) * np.array([1, 1, 0.3])).T).T).astype(F)
        lo, hi = (c - hs).astype(F), (c + hs).astype(F)
        gd = _plane_guard(fx, fy, cols, rows, ti, lo, hi)
        right = np.ones(nb, bool); left_c = np.ones(nb, bool); left_s = np.ones(nb, bool); below = np.ones(nb, bool); above = np.ones(nb, bool)
        for k in range(8):
            w = np.where(np.array([(k >> a) & 1 for a in range(3)], bool), hi, lo).astype(F)
            p = np.stack([((Ri[r, 0] * w[:, 0] + Ri[r, 1] * w[:, 1]).astype(F) + Ri[r, 2] * w[:, 2]).astype(F) + ti[r] for r in range(3)], 1).astype(F)
            right &= (fx * p[:, 0] + (cx - cols - F(2.0)) * p[:, 2]).astype(F) > gd
            left_c &= (fx * p[:, 0] + (cx - border + F(2.0)) * p[:, 2]).astype(F) < -gd
            left_s &= (fx * p[:, 0] + (cx + F(2.0)) * p[:, 2]).astype(F) < -gd
            below &= (fy * p[:, 1] + (cy - rows - F(2.0)) * p[:, 2]).astype(F) > gd
            above &= (fy * p[:, 1] + (cy + F(2.0)) * p[:, 2]).astype(F) < -gd
        cull_conf = right | left_c | below | above
        cull_splat = right | left_s | below | above
        culled_any += int(cull_splat.sum())
        # sample points of every box (corners, edge and interior points), transform as the kernels do, apply the exact view tests
        u01 = np.concatenate([rng.random((56, 3)), np.array([[(k >> a) & 1 for a in range(3)] for k in range(8)], float)]).astype(F)
        for s in u01:
            w = (lo + (hi - lo) * s).astype(F)
            p = np.stack([((Ri[r, 0] * w[:, 0] + Ri[r, 1] * w[:, 1]).astype(F) + Ri[r, 2] * w[:, 2]).astype(F) + ti[r] for r in range(3)], 1).astype(F)
            z = p[:, 2]
            with np.errstate(all="ignore"):
                xl = (p[:, 0] / z).astype(F); yl = (p[:, 1] / z).astype(F)
                u = (fx * xl + cx).astype(F); v = (fy * yl + cy).astype(F)
                in_conf = (z > F(1.0)) & (z < F(30.0)) & ~((u < border) | (u > cols) | (v < 0) | (v > rows))             # conflict.vert:25-49
                us = (((fx * p[:, 0]).astype(F) / z).astype(F) + cx).astype(F); vs = (((fy * p[:, 1]).astype(F) / z).astype(F) + cy).astype(F)
                in_splat = (z > 0) & (z < F(45.0)) & (us >= 0) & (us <= cols) & (vs >= 0) & (vs <= rows)                 # index_map.vert (superset of its tests)
            assert not np.any(in_conf & cull_conf), trial
            assert not np.any(in_splat & cull_splat), trial
    assert culled_any > 10_000            # the test is not vacuous: most boxes beside / behind the camera are culled


def test_pass_pretest_is_a_superset_of_both_exact_view_tests():
    """k_surfel_pass phase A (pass_tile_compact): x * rcp(z) with a 2-pixel margin instead of the correctly rounded quotients;
    rcp taken 1 ulp low / exact / 1 ulp high (v_rcp_f32 is a 1-ulp instruction).  No lane the exact tests accept may be dropped."""
    rng = np.random.default_rng(11)
    fx, fy, cx, cy, cols, rows = F(718.856), F(718.856), F(607.19), F(185.2), F(1242.0), F(375.0)
    border, zmin, zmax, cutoff = F(80.0), F(1.0), F(30.0), F(30.0)
    n = 3_000_000
    z = np.concatenate([rng.uniform(-1, 50, n // 2), np.abs(rng.normal(0, 0.05, n // 4)) + 1e-6, rng.uniform(0.9, 1.1, n // 4)]).astype(F)
    # image coordinates concentrated around the four borders (+- a few pixels) and spread over / beyond the image
    ut = np.where(rng.random(n) < 0.5, rng.choice([0.0, float(border), float(cols)], n) + rng.normal(0, 1.5, n), rng.uniform(-200, float(cols) + 200, n))
    vt = np.where(rng.random(n) < 0.5, rng.choice([0.0, float(rows)], n) + rng.normal(0, 1.5, n), rng.uniform(-100, float(rows) + 100, n))
    x = ((ut - float(cx)) / float(fx) * z).astype(F)
    y = ((vt - float(cy)) / float(fy) * z).astype(F)
    with np.errstate(all="ignore"):
        xl = (x / z).astype(F); yl = (y / z).astype(F)
        u = (fx * xl + cx).astype(F); v = (fy * yl + cy).astype(F)
        in_conf = ~((z <= zmin) | (z >= zmax)) & ~((u < border) | (u > cols) | (v < 0) | (v > rows))
        hc, hr = (cols * F(0.5)).astype(F), (rows * F(0.5)).astype(F)
        xn = ((((((fx * x).astype(F) / z).astype(F) + cx).astype(F)) - hc).astype(F) / hc).astype(F)
        yn = ((((((fy * y).astype(F) / z).astype(F) + cy).astype(F)) - hr).astype(F) / hr).astype(F)
        zn = (z / cutoff).astype(F)
        in_splat = ~((z >= (cutoff * F(1.5)).astype(F)) | (z <= 0)) & (xn >= -1) & (xn <= 1) & (yn >= -1) & (yn <= 1) & (zn >= -1) & (zn <= 1)
        r0 = (F(1.0) / z).astype(F)
        for r in (np.nextafter(r0, F(-np.inf)), r0, np.nextafter(r0, F(np.inf))):
            ua = ((fx * x).astype(F) * r + cx).astype(F); va = ((fy * y).astype(F) * r + cy).astype(F)
            out_img = (ua < F(-2.0)) | (ua > cols + F(2.0)) | (va < F(-2.0)) | (va > rows + F(2.0))
            rej_c = (z <= zmin) | (z >= zmax) | out_img
            rej_s = (z >= (cutoff * F(1.5)).astype(F)) | (z <= 0) | out_img
            assert not np.any(in_conf & rej_c)
            assert not np.any(in_splat & rej_s)
    assert in_conf.mean() > 0.05 and in_splat.mean() > 0.05 and (rej_c & rej_s).mean() > 0.2
