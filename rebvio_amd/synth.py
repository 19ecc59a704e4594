"""Deterministic synthetic camera streams for tests and bench.

Scene: a textured fronto-parallel plane at Z=2 m with smooth-edged holes (about 30 % of the
pixels) through which a second textured plane at Z=4 m is seen. Texture = random convex polygons
of random contrast plus a few spatially modulated gratings, so that the keyline count falls off
smoothly with the detector threshold (the threshold servo of EdgeDetector::detect then settles at
`keylines_ref`). Camera: pinhole (fm, cx, cy), no distortion; constant body velocity plus a slow
yaw. Frames are u8; the pipeline input is u8*3.0f as after `convertTo(CV_32F, 3.0)`
(reference rebvio.cpp:43).

This module is plumbing (numpy only): it produces inputs, it computes nothing of the hot path.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np
from scipy.ndimage import gaussian_filter

SEED = 0x5EB10


@dataclasses.dataclass
class Camera:
    width: int
    height: int
    fm: float
    cx: float
    cy: float
    dist: tuple = (0.0, 0.0, 0.0, 0.0, 0.0)   # rad-tan k1, k2, p1, p2, k3 (camera.hpp:31-35); zero = pinhole

    def undistorted_rays(self):
        """Normalised pinhole coordinates (x, y) seen by every pixel of the (possibly distorting) lens: inverts
        xd = x*kr + 2 p1 x y + p2 (r2 + 2 x^2), yd = y*kr + p1 (r2 + 2 y^2) + 2 p2 x y by fixed-point iteration."""
        xd, yd = np.meshgrid((np.arange(self.width, dtype=np.float64) - self.cx) / self.fm,
                             (np.arange(self.height, dtype=np.float64) - self.cy) / self.fm)
        k1, k2, p1, p2, k3 = self.dist
        if not any(self.dist):
            return xd, yd
        x, y = xd.copy(), yd.copy()
        for _ in range(20):
            r2 = x * x + y * y
            kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x, y = (xd - dx) / kr, (yd - dy) / kr
        return x, y

    @staticmethod
    def for_size(width: int, height: int) -> "Camera":
        # fm = 458 at 640x480, scaled with resolution (x2 for 1280x960)
        s = width / 640.0
        return Camera(width, height, 458.0 * s, width / 2.0, height / 2.0)


@dataclasses.dataclass
class Scene:
    poly_n: np.ndarray      # [K, E, 2] edge normals (front plane)
    poly_d: np.ndarray      # [K, E]
    poly_amp: np.ndarray    # [K]
    poly_bb: np.ndarray     # [K, 4] umin, umax, vmin, vmax
    grat: np.ndarray        # [G, 6] kx, ky, phase, amp, env_kx, env_ky
    hole_c: np.ndarray      # [H, 3] cu, cv, radius
    back_poly_n: np.ndarray
    back_poly_d: np.ndarray
    back_poly_amp: np.ndarray
    back_poly_bb: np.ndarray
    back_grat: np.ndarray
    z_front: float = 2.0
    z_back: float = 4.0
    vel: tuple = (0.02, 0.005, 0.01)   # m / frame
    yaw_deg: float = 0.2               # deg / frame about y


def _polys(rng, k, extent_u, extent_v, size_lo, size_hi, amp_lo, amp_hi):
    n_edges = 4
    cu = rng.uniform(-extent_u, extent_u, k)
    cv = rng.uniform(-extent_v, extent_v, k)
    rad = rng.uniform(size_lo, size_hi, k)
    base = rng.uniform(0, 2 * math.pi, k)
    normals = np.zeros((k, n_edges, 2), np.float32)
    dist = np.zeros((k, n_edges), np.float32)
    for e in range(n_edges):
        ang = base + e * (2 * math.pi / n_edges) + rng.uniform(-0.35, 0.35, k)
        nx, ny = np.cos(ang), np.sin(ang)
        r = rad * rng.uniform(0.6, 1.0, k)
        normals[:, e, 0] = nx
        normals[:, e, 1] = ny
        dist[:, e] = nx * cu + ny * cv + r
    amp = rng.uniform(amp_lo, amp_hi, k) * rng.choice([-1.0, 1.0], k)
    bb = np.stack([cu - 1.5 * rad, cu + 1.5 * rad, cv - 1.5 * rad, cv + 1.5 * rad], 1)
    return normals, dist, amp.astype(np.float32), bb.astype(np.float32)


def _gratings(rng, g, period_lo, period_hi, amp_lo, amp_hi):
    out = np.zeros((g, 6), np.float32)
    for i in range(g):
        ang = rng.uniform(0, math.pi)
        k = 2 * math.pi / rng.uniform(period_lo, period_hi)
        eang = rng.uniform(0, math.pi)
        ek = 2 * math.pi / rng.uniform(1.5, 3.0)
        out[i] = (k * math.cos(ang), k * math.sin(ang), rng.uniform(0, 2 * math.pi), rng.uniform(amp_lo, amp_hi),
                  ek * math.cos(eang), ek * math.sin(eang))
    return out


def make_scene(stream_id: int = 0, density: float = 1.0) -> Scene:
    rng = np.random.Generator(np.random.PCG64(SEED + stream_id))
    eu, ev = 3.2, 2.4
    kf = int(150 * density)
    pn, pd, pa, pb = _polys(rng, kf, eu, ev, 0.10, 0.45, 4.0, 70.0)
    gr = _gratings(rng, 3, 0.09, 0.16, 4.0, 18.0)
    nh = 14
    holes = np.stack([rng.uniform(-eu, eu, nh), rng.uniform(-ev, ev, nh), rng.uniform(0.35, 0.6, nh)], 1).astype(np.float32)
    kb = int(170 * density)
    bn, bd, ba, bbb = _polys(rng, kb, 2.2 * eu, 2.2 * ev, 0.2, 0.9, 4.0, 70.0)
    bg = _gratings(rng, 2, 0.2, 0.35, 4.0, 16.0)
    return Scene(pn, pd, pa, pb, gr, holes, bn, bd, ba, bbb, bg)


def _texture(u, v, normals, dist, amp, bb, grat, soft):
    """Sum of soft convex polygons + modulated gratings at plane coordinates (u, v)."""
    val = np.full(u.shape, 118.0, np.float32)
    umin, umax, vmin, vmax = float(u.min()), float(u.max()), float(v.min()), float(v.max())
    inv = np.float32(1.0 / soft)
    for k in range(normals.shape[0]):
        b = bb[k]
        if b[1] < umin or b[0] > umax or b[3] < vmin or b[2] > vmax:
            continue
        m = None
        for e in range(normals.shape[1]):
            s = (dist[k, e] - normals[k, e, 0] * u - normals[k, e, 1] * v) * inv
            np.clip(s, -0.5, 0.5, out=s)
            s += 0.5
            m = s if m is None else np.minimum(m, s, out=m)
        val += amp[k] * m
    for g in grat:
        env = 0.5 + 0.5 * np.sin(g[4] * u + g[5] * v)
        val += g[3] * env * np.sin(g[0] * u + g[1] * v + g[2])
    return val


def pose(scene: Scene, t: float):
    """Camera-to-world rotation and position at (fractional) frame index t."""
    a = math.radians(scene.yaw_deg * t)
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], np.float64)
    p = np.array(scene.vel, np.float64) * t
    return R, p


def render_frame(scene: Scene, cam: Camera, t: float, noise_seed: int | None = None, noise_sigma: float = 2.0) -> np.ndarray:
    """Render frame t as u8 [height, width]."""
    R, p = pose(scene, t)
    X, Y = cam.undistorted_rays()
    dx = R[0, 0] * X + R[0, 1] * Y + R[0, 2]
    dy = R[1, 0] * X + R[1, 1] * Y + R[1, 2]
    dz = R[2, 0] * X + R[2, 1] * Y + R[2, 2]
    px_m = scene.z_front / cam.fm  # one pixel in metres on the front plane

    def plane_uv(z):
        s = (z - p[2]) / dz
        return (p[0] + s * dx).astype(np.float32), (p[1] + s * dy).astype(np.float32)

    uf, vf = plane_uv(scene.z_front)
    front = _texture(uf, vf, scene.poly_n, scene.poly_d, scene.poly_amp, scene.poly_bb, scene.grat, 1.2 * px_m)
    # soft hole mask on the front plane
    hole = np.zeros(uf.shape, np.float32)
    for cu, cv, r in scene.hole_c:
        d = np.sqrt((uf - cu) ** 2 + (vf - cv) ** 2)
        hole = np.maximum(hole, np.clip((r - d) / (1.2 * px_m) + 0.5, 0.0, 1.0))
    ub, vb = plane_uv(scene.z_back)
    back = _texture(ub, vb, scene.back_poly_n, scene.back_poly_d, scene.back_poly_amp, scene.back_poly_bb,
                    scene.back_grat, 1.2 * 2 * px_m)
    img = front * (1.0 - hole) + (back - 18.0) * hole
    img = gaussian_filter(img, 1.0, mode="nearest")
    if noise_seed is not None and noise_sigma > 0:
        nrng = np.random.Generator(np.random.PCG64(SEED * 7919 + noise_seed))
        img = img + nrng.normal(0.0, noise_sigma, img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def render_stream(width: int, height: int, n_frames: int, stream_id: int = 0, density: float = 1.0, noise: bool = True,
                  dist=None):
    """Frames 0..n_frames-1 of stream `stream_id` -> (u8 [n, H, W], Camera). `dist` = rad-tan lens (k1,k2,p1,p2,k3): the
    frames are then what that lens sees, and undistorting them gives back the pinhole view."""
    cam = Camera.for_size(width, height)
    if dist is not None:
        cam.dist = tuple(float(v) for v in dist)
    scene = make_scene(stream_id, density)
    # keep metric motion per pixel constant across resolutions
    frames = np.empty((n_frames, height, width), np.uint8)
    for t in range(n_frames):
        frames[t] = render_frame(scene, cam, float(t), noise_seed=(stream_id * 100003 + t) if noise else None)
    return frames, cam


def imu_samples(scene: Scene, n_frames: int, frame_dt_us: int = 50000, rate_hz: int = 200, R_c2i=None, noise_seed: int | None = None,
                gyro_bias=(0.0, 0.0, 0.0)):
    """Analytic IMU stream of the scene's motion (BASELINE config 5): constant world velocity (zero acceleration) and a
    constant yaw rate about the camera's y axis, gravity along +y of the camera (image "down").
    Returns (ts_us[int64], gyro[n,3], acc[n,3]) in the IMU frame; sample k of frame f has f*dt < ts <= (f+1)*dt... i.e.
    timestamps in (0, (n_frames-1)*dt]."""
    R_c2i = np.eye(3) if R_c2i is None else np.asarray(R_c2i, np.float64)
    step = 1000000 // rate_hz  # the IMU runs on its own clock: 5000 us at 200 Hz whatever the camera rate
    ts = np.arange(1, ((n_frames - 1) * frame_dt_us) // step + 1, dtype=np.int64) * step
    yaw_rate = math.radians(scene.yaw_deg) * (1e6 / frame_dt_us)  # rad/s about camera y
    w_cam = np.array([0.0, yaw_rate, 0.0])
    f_cam = np.array([0.0, -9.81, 0.0])  # specific force = a - g with a = 0, g = +9.81 along camera y (invariant under yaw)
    gyro = np.tile(R_c2i @ w_cam + np.asarray(gyro_bias), (len(ts), 1))
    acc = np.tile(R_c2i @ f_cam, (len(ts), 1))
    if noise_seed is not None:
        rng = np.random.Generator(np.random.PCG64(SEED * 31 + noise_seed))
        gyro = gyro + rng.normal(0, 1.6968e-4, gyro.shape)
        acc = acc + rng.normal(0, 2.0e-3, acc.shape)
    return ts, gyro.astype(np.float32), acc.astype(np.float32)


def pingpong_indices(n_base: int, n_total: int) -> np.ndarray:
    """0,1,..,n-1,n-2,..,1,0,1,.. : every consecutive pair is a valid small inter-frame motion."""
    period = list(range(n_base)) + list(range(n_base - 2, 0, -1))
    reps = (n_total + len(period) - 1) // len(period)
    return np.array((period * reps)[:n_total], np.int32)
