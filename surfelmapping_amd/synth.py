"""Deterministic synthetic RGB-D + semantic frames.

No dataset is available offline (the reference's demo data is a Google-Drive link,
README.md:28), so every test and the benchmark use frames ray-cast from a small analytic
street scene, shaped like the reference's inputs (gui/KittiReader.cpp:56-305):

    rgb      u8 [H][W][3]  (R first)
    depth    u16[H][W]     millimetres, 0 = invalid
    semantic u8 [H][W]     Cityscapes train ids (0 road, 2 building, 10 sky, 13 car)
    pose     f32[16]       column-major camera->world

Camera frame is KITTI's: x right, y down, z forward.  Scene: ground plane y = +1.65 m,
two walls x = +-8 m (up to 3 m above the camera), sky above, and a few car-sized boxes.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

KITTI = dict(width=1242, height=375, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157)
VGA = dict(width=640, height=480, fx=525.0, fy=525.0, cx=319.5, cy=239.5)
HD = dict(width=1920, height=1080, fx=1000.0, fy=1000.0, cx=960.0, cy=540.0)


@dataclasses.dataclass
class Camera:
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float


def _hash_u32(a: np.ndarray) -> np.ndarray:
    a = a.astype(np.uint64) & 0xFFFFFFFF
    a = (a ^ (a >> 16)) * 0x45D9F3B & 0xFFFFFFFF
    a = (a ^ (a >> 16)) * 0x45D9F3B & 0xFFFFFFFF
    a = a ^ (a >> 16)
    return a.astype(np.uint32)


class Scene:
    """Analytic street scene; `seed` fixes the boxes."""

    def __init__(self, seed: int = 0, n_boxes: int = 6, length: float = 200.0):
        rng = np.random.default_rng(0x5EED0000 + seed)
        self.ground_y = 1.65
        self.wall_x = 8.0
        self.wall_top = -3.0
        boxes = []
        for _ in range(n_boxes):
            cx = rng.uniform(-5.5, 5.5)
            cz = rng.uniform(8.0, length)
            sx, sy, sz = rng.uniform(1.6, 2.0), rng.uniform(1.4, 1.8), rng.uniform(3.5, 4.5)
            boxes.append((cx - sx / 2, self.ground_y - sy, cz - sz / 2,
                          cx + sx / 2, self.ground_y, cz + sz / 2))
        self.boxes = np.array(boxes, dtype=np.float64).reshape(-1, 6)

    def render(self, cam: Camera, pose: np.ndarray, noise_mm: float = 0.0, noise_seed: int = 0):
        """Ray-cast one frame. `pose` is a 4x4 camera->world matrix (row/col indexable)."""
        W, H = cam.width, cam.height
        i = np.arange(W, dtype=np.float64) + 0.5
        j = np.arange(H, dtype=np.float64) + 0.5
        dx = (i - cam.cx) / cam.fx
        dy = (j - cam.cy) / cam.fy
        dirs_c = np.stack(np.broadcast_arrays(dx[None, :], dy[:, None], np.ones((H, W))), axis=-1)
        R = np.asarray(pose, dtype=np.float64)[:3, :3]
        o = np.asarray(pose, dtype=np.float64)[:3, 3]
        d = dirs_c @ R.T                                   # world directions, param = camera z
        best = np.full((H, W), np.inf)
        sid = np.zeros((H, W), dtype=np.int32)             # 0 none/sky
        cls = np.full((H, W), 10, dtype=np.uint8)

        def consider(t, mask, surface_id, klass):
            nonlocal best, sid, cls
            m = mask & (t > 1e-6) & (t < best)
            best = np.where(m, t, best)
            sid = np.where(m, surface_id, sid)
            cls = np.where(m, klass, cls)

        with np.errstate(divide="ignore", invalid="ignore"):
            # ground plane y = ground_y
            t = (self.ground_y - o[1]) / d[..., 1]
            consider(t, np.isfinite(t), 1, 0)
            # walls x = +-wall_x, limited in height
            for s, surf in ((-1.0, 2), (1.0, 3)):
                t = (s * self.wall_x - o[0]) / d[..., 0]
                y = o[1] + t * d[..., 1]
                consider(t, np.isfinite(t) & (y >= self.wall_top) & (y <= self.ground_y), surf, 2)
            # boxes (slab test)
            for b, box in enumerate(self.boxes):
                lo, hi = box[:3], box[3:]
                t0 = (lo - o) / d
                t1 = (hi - o) / d
                tn = np.nanmax(np.minimum(t0, t1), axis=-1)
                tf = np.nanmin(np.maximum(t0, t1), axis=-1)
                consider(tn, (tn <= tf) & np.isfinite(tn), 10 + b, 13)

        hit = np.isfinite(best)
        z = np.where(hit, best, 0.0)
        if noise_mm > 0.0:
            rng = np.random.default_rng(0xD00D0000 + noise_seed)
            z = z + np.where(hit, rng.normal(0.0, noise_mm * 1e-3, size=z.shape), 0.0)
        mm = np.clip(np.rint(z * 1000.0), 0, 65535)
        depth = np.where(hit, mm, 0).astype(np.uint16)
        # colour: hash of the surface id and a 0.25 m world grid cell
        pw = o[None, None, :] + np.where(hit, best, 0.0)[..., None] * d
        cell = np.floor(pw * 4.0).astype(np.int64)
        h = _hash_u32((cell[..., 0] * 73856093) ^ (cell[..., 1] * 19349663) ^
                      (cell[..., 2] * 83492791) ^ (sid.astype(np.int64) * 2654435761))
        rgb = np.stack([(h & 0xFF), (h >> 8) & 0xFF, (h >> 16) & 0xFF], axis=-1).astype(np.uint8)
        sky = np.array([135, 206, 235], dtype=np.uint8)
        rgb = np.where(hit[..., None], rgb, sky[None, None, :]).astype(np.uint8)
        return (np.ascontiguousarray(rgb), np.ascontiguousarray(depth),
                np.ascontiguousarray(cls.astype(np.uint8)))


def pose_matrix(tx: float, ty: float, tz: float, yaw_deg: float = 0.0) -> np.ndarray:
    """camera->world 4x4 (numpy row/col indexed), yaw about the camera's y (down) axis."""
    a = math.radians(yaw_deg)
    c, s = math.cos(a), math.sin(a)
    m = np.eye(4, dtype=np.float64)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    m[:3, 3] = (tx, ty, tz)
    return m


def pose_to_colmajor(m: np.ndarray) -> np.ndarray:
    """4x4 numpy matrix -> float32[16] column-major (Eigen::Matrix4f storage)."""
    return np.ascontiguousarray(np.asarray(m, dtype=np.float32).T.reshape(16))


def kitti_trajectory(n_frames: int, step: float = 0.8):
    """Forward motion `step` m/frame along +z with yaw 0.5deg*sin(k/20)."""
    return [pose_matrix(0.0, 0.0, step * k, 0.5 * math.sin(k / 20.0)) for k in range(n_frames)]


def make_sequence(cam_kw: dict, poses, seed: int = 0, noise_mm: float = 0.0, scene: Scene | None = None):
    """Render a list of frames: [(rgb, depth, sem, pose16_colmajor), ...]."""
    cam = Camera(**cam_kw)
    scene = scene or Scene(seed)
    out = []
    for k, p in enumerate(poses):
        rgb, depth, sem = scene.render(cam, p, noise_mm=noise_mm, noise_seed=seed * 100003 + k)
        out.append((rgb, depth, sem, pose_to_colmajor(p)))
    return out


def _render_job(job):
    cam_kw, pose, seed, k, noise_mm, scene_kw = job
    scene = Scene(**scene_kw) if scene_kw else Scene(seed)
    rgb, depth, sem = scene.render(Camera(**cam_kw), pose, noise_mm=noise_mm, noise_seed=seed * 100003 + k)
    return rgb, depth, sem, pose_to_colmajor(pose)


def make_sequences_parallel(specs, workers: int = 8):
    """Several sequences at once on `workers` fresh processes (spawned, not forked: safe in a process that holds a GPU context).
    specs: [(cam_kw, poses, seed, noise_mm, scene_kw or None), ...] with scene_kw the arguments of Scene (None: Scene(seed));
    returns the sequences in the same order, each as make_sequence would."""
    import multiprocessing as mp
    jobs, owner = [], []
    for si, (cam_kw, poses, seed, noise_mm, scene_kw) in enumerate(specs):
        for k, p in enumerate(poses):
            jobs.append((cam_kw, p, seed, k, noise_mm, scene_kw))
            owner.append(si)
    if workers <= 1 or len(jobs) <= 1:
        res = [_render_job(j) for j in jobs]
    else:
        with mp.get_context("spawn").Pool(min(workers, len(jobs))) as pool:
            res = pool.map(_render_job, jobs, chunksize=1)
    out = [[] for _ in specs]
    for si, r in zip(owner, res):
        out[si].append(r)
    return out


def seeded_model(n: int, tick: int, seed: int = 0) -> np.ndarray:
    """Config-3 style pre-seeded model: n surfels, AoS float32[n][12]."""
    rng = np.random.default_rng(0xABCD0000 + seed)
    m = np.zeros((n, 12), dtype=np.float32)
    m[:, 0] = rng.uniform(-60, 60, n)
    m[:, 1] = rng.uniform(-3, 5, n)
    m[:, 2] = rng.uniform(-50, 250, n)
    m[:, 3] = rng.choice(np.array([0.9, 1.8, 2.7], dtype=np.float32), n)
    sem = rng.integers(0, 19, n, dtype=np.uint32)
    col = rng.integers(0, 1 << 24, n, dtype=np.uint32)
    m[:, 4] = ((sem << 24) | col).view(np.float32)
    t = (tick - rng.integers(0, 301, n)).astype(np.float32)
    m[:, 6] = t
    m[:, 7] = t
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    m[:, 8:11] = v.astype(np.float32)
    m[:, 11] = rng.uniform(0.02, 0.1, n)
    return m
