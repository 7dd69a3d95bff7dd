"""Synthetic depth stream of SURVEY.md section 8(d): analytic box room + one sphere.

This is the input generator shared by the parity tests and ``bench.py`` (BASELINE.json
configs 2, 4 and 5).  It is not part of the hot path and touches neither the oracle nor
the HIP library.

Scene (all lengths as fractions of the volume edge ``dim``):
  * axis-aligned box room ``[0.05, 0.95]^3`` seen from inside,
  * sphere centred at ``(0.5, 0.5, 0.62)``, radius ``0.08``.
Camera: ``k = (481.2, 480, 320, 240) * (W / 640)`` (the ICL-NUIM intrinsics of the reference's
README.md:80 with positive fy), pose(f) = translation ``(0.34, 0.5, 0.24) * dim + (1, 0, 0.5) mm * f``,
rotation = yaw of ``0.2 deg * f`` about +y, camera looking along +z.
Depth = z-depth of the nearest surface along the ray through the pixel centre, truncated to
uint16 millimetres and divided by 1000 (what the reference's readers + ``mm2metersKernel``
deliver: se_apps/include/interface.h:227-236, se_denseslam/src/preprocessing.cpp:184-185);
2 % of the pixels are zeroed with one ``std::mt19937(54321)`` stream for the whole run
(``uniform_real_distribution<float>(0,1) < 0.02`` per pixel in row-major order).
"""
from __future__ import annotations

import numpy as np

ROOM_LO, ROOM_HI = 0.05, 0.95
SPHERE_C = (0.5, 0.5, 0.62)
SPHERE_R = 0.08
HOLE_SEED = 54321
HOLE_FRACTION = np.float32(0.02)


def intrinsics(width: int, negative_fy: bool = False) -> np.ndarray:
    """(fx, fy, cx, cy) scaled from the 640-wide ICL-NUIM camera.  ``negative_fy``: the ICL-NUIM
    convention proper (``-k 481.2,-480,320,240``, README.md:80 of the reference): image rows run
    against the camera's +y, which is what BASELINE.json configs 1 and 3 use."""
    fy = -480.0 if negative_fy else 480.0
    return (np.array([481.2, fy, 320.0, 240.0], dtype=np.float64) * (width / 640.0)).astype(np.float32)


def pose(frame: int, dim: float) -> np.ndarray:
    """Camera->world 4x4 float32 matrix for ``frame`` (row-major numpy array)."""
    a = np.deg2rad(0.2 * frame)
    c, s = np.cos(a), np.sin(a)
    T = np.eye(4, dtype=np.float64)
    T[:3, :3] = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
    T[:3, 3] = np.array([0.34, 0.5, 0.24]) * dim + np.array([1.0, 0.0, 0.5]) * 1e-3 * frame
    return T.astype(np.float32)


def to_colmajor(m: np.ndarray) -> np.ndarray:
    """16 floats in the column-major order of ``Eigen::Matrix4f::data()`` (the C-ABI layout)."""
    return np.ascontiguousarray(np.asarray(m, dtype=np.float32).T).reshape(16)


class HoleStream:
    """libstdc++ ``std::mt19937(seed)`` + ``uniform_real_distribution<float>(0,1)`` replica."""

    def __init__(self, seed: int = HOLE_SEED):
        self._bg = np.random.MT19937()
        self._bg._legacy_seeding(seed)  # init_genrand(seed), as std::mt19937(seed)

    def uniform(self, n: int) -> np.ndarray:
        raw = self._bg.random_raw(n).astype(np.uint32)
        v = raw.astype(np.float32) / np.float32(4294967296.0)
        return np.where(v >= 1, np.nextafter(np.float32(1), np.float32(0)), v).astype(np.float32)


def render_depth_mm(frame: int, width: int, height: int, dim: float, negative_fy: bool = False) -> np.ndarray:
    """uint16 millimetre depth image of the analytic scene (no holes)."""
    k = intrinsics(width, negative_fy).astype(np.float64)
    T = pose(frame, dim).astype(np.float64)
    xs = (np.arange(width) + 0.5 - k[2]) / k[0]
    ys = (np.arange(height) + 0.5 - k[3]) / k[1]
    u, v = np.meshgrid(xs, ys)
    d_cam = np.stack([u, v, np.ones_like(u)], axis=-1)
    d = d_cam @ T[:3, :3].T          # world direction, parametrised so that t == camera z-depth
    o = T[:3, 3]
    lo, hi = ROOM_LO * dim, ROOM_HI * dim
    with np.errstate(divide="ignore", invalid="ignore"):
        t_axis = np.where(d > 0, (hi - o) / d, np.where(d < 0, (lo - o) / d, np.inf))
    t_room = t_axis.min(axis=-1)
    c = np.array(SPHERE_C) * dim
    r = SPHERE_R * dim
    oc = o - c
    A = (d * d).sum(-1)
    B = 2.0 * (d * oc).sum(-1)
    C = (oc * oc).sum() - r * r
    disc = B * B - 4 * A * C
    sq = np.sqrt(np.maximum(disc, 0.0))
    t_s = (-B - sq) / (2 * A)
    t_s = np.where((disc >= 0) & (t_s > 0), t_s, np.inf)
    depth = np.minimum(t_room, t_s)
    mm = np.floor(depth * 1000.0)
    return np.clip(mm, 0, 65535).astype(np.uint16)


class SyntheticStream:
    """Iterator-free frame source: ``depth(f)`` must be called for f = 0, 1, 2, ... in order
    (the hole stream is one RNG for the whole run, as in the survey's probe)."""

    def __init__(self, width: int, height: int, dim: float, holes: bool = True, negative_fy: bool = False):
        self.width, self.height, self.dim = width, height, float(dim)
        self.negative_fy = negative_fy
        self.k = intrinsics(width, negative_fy)
        self._holes = HoleStream() if holes else None
        self._next = 0

    def depth(self, frame: int) -> np.ndarray:
        if frame != self._next:
            raise ValueError("SyntheticStream frames must be requested in order")
        self._next += 1
        mm = render_depth_mm(frame, self.width, self.height, self.dim, self.negative_fy)
        d = mm.astype(np.float32) / np.float32(1000.0)   # mm2metersKernel: depth / 1000.0f
        if self._holes is not None:
            u = self._holes.uniform(self.width * self.height).reshape(self.height, self.width)
            d = np.where(u < HOLE_FRACTION, np.float32(0), d)
        return np.ascontiguousarray(d, dtype=np.float32)

    def pose(self, frame: int) -> np.ndarray:
        return pose(frame, self.dim)


def surface_distance(points: np.ndarray, dim: float) -> np.ndarray:
    """Distance of world points to the analytic room / sphere surface (chaos-robust metric of
    SURVEY.md section 8(d))."""
    p = np.asarray(points, dtype=np.float64)
    lo, hi = ROOM_LO * dim, ROOM_HI * dim
    d_wall = np.minimum(np.abs(p - lo), np.abs(hi - p)).min(axis=-1)
    d_sph = np.abs(np.linalg.norm(p - np.array(SPHERE_C) * dim, axis=-1) - SPHERE_R * dim)
    return np.minimum(d_wall, d_sph)


# ------------------------------------------------------------------------------------------------
# "ICL-like stress" stream: the regime a real living_room_traj2 run lives in and the box-room
# stream above never enters (missing #3):
#   * the room is LARGER than the volume on two sides (the -x wall and the +z wall lie outside the
#     cube, floor and ceiling run through its x = 0 and z = dim faces): depth points and allocation
#     band steps outside the volume (kfusion/alloc_impl.hpp:92-96 skips them), surfaces clipped by
#     the cube, rays whose cube exit or far plane ends them (ray_iterator.hpp:95-102), interpolation
#     and gradient stencils at the volume faces (octree.hpp:541-563, 652-737);
#   * a pillar in front of the walls and a second sphere that straddles the x = 0 face: depth
#     discontinuities, curvature cut by the cube;
#   * centimetre-scale motion: ~1.3 cm and 2 deg per frame, yaw as a triangular wave of +-90 deg (a
#     180 deg pan there and back: blocks leave the frustum -> active(false),
#     projective_functor.hpp:110, and come back), a small pitch oscillation on top;
#   * depths up to ~4.6 m, i.e. beyond farPlane = 4.0 m (constant_parameters.h:32);
#   * SURVEY 8(d)'s sensor noise: sigma = 1 mm Gaussian from one mt19937(12345) stream for the run
#     (Box-Muller on the raw 32-bit words in float64, row-major, two words per pixel), added before the
#     truncation to uint16 millimetres; plus the 2 % zero-depth holes of the stream above.
# All lengths are fractions of the volume edge, as above.
# ------------------------------------------------------------------------------------------------
STRESS_ROOM_LO = (-0.12, 0.06, 0.04)
STRESS_ROOM_HI = (0.93, 0.94, 1.18)
STRESS_SPHERES = (((0.5, 0.5, 0.62), 0.08), ((0.02, 0.40, 0.50), 0.07))
STRESS_PILLAR = ((0.60, 0.06, 0.36), (0.70, 0.80, 0.44))
NOISE_SEED = 12345
NOISE_SIGMA_MM = 1.0


def stress_pose(frame: int, dim: float) -> np.ndarray:
    """Camera->world pose of the stress stream: a slow loop (radius 0.25 m, period 120 frames) around the ICL
    start position, yaw = triangular wave of amplitude 90 deg at 2 deg / frame, pitch = 5 deg * sin(2 pi f / 50)."""
    ph = frame % 180
    yaw_deg = 2.0 * ph if ph <= 45 else (2.0 * (90 - ph) if ph <= 135 else 2.0 * (ph - 180))
    yaw, pitch = np.deg2rad(yaw_deg), np.deg2rad(5.0 * np.sin(2 * np.pi * frame / 50.0))
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    w = 2 * np.pi * frame / 120.0
    T = np.eye(4, dtype=np.float64)
    T[:3, :3] = Ry @ Rx
    T[:3, 3] = np.array([0.34, 0.5, 0.24]) * dim + np.array([0.25 * np.sin(w), 0.05 * np.sin(3 * w), 0.25 * (1 - np.cos(w))])
    return T.astype(np.float32)


def _ray_box_outside(o, d, lo, hi):
    """Entry distance of rays (o + t d) into the axis-aligned box [lo, hi] seen from outside; inf if missed."""
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (lo - o) / d
        t1 = (hi - o) / d
    tn = np.where(d != 0, np.minimum(t0, t1), np.where((o >= lo) & (o <= hi), -np.inf, np.inf))
    tf = np.where(d != 0, np.maximum(t0, t1), np.where((o >= lo) & (o <= hi), np.inf, -np.inf))
    tn, tf = tn.max(axis=-1), tf.min(axis=-1)
    return np.where((tn <= tf) & (tn > 0), tn, np.inf)


def _ray_sphere(o, d, c, r):
    oc = o - c
    A = (d * d).sum(-1)
    B = 2.0 * (d * oc).sum(-1)
    Cc = (oc * oc).sum() - r * r
    disc = B * B - 4 * A * Cc
    t = (-B - np.sqrt(np.maximum(disc, 0.0))) / (2 * A)
    return np.where((disc >= 0) & (t > 0), t, np.inf)


def render_stress_depth(frame: int, width: int, height: int, dim: float, negative_fy: bool = True) -> np.ndarray:
    """float64 z-depth in metres of the stress scene (no noise, no holes, not yet quantised)."""
    k = intrinsics(width, negative_fy).astype(np.float64)
    T = stress_pose(frame, dim).astype(np.float64)
    xs = (np.arange(width) + 0.5 - k[2]) / k[0]
    ys = (np.arange(height) + 0.5 - k[3]) / k[1]
    u, v = np.meshgrid(xs, ys)
    d = np.stack([u, v, np.ones_like(u)], axis=-1) @ T[:3, :3].T   # t == camera z-depth
    o = T[:3, 3]
    lo, hi = np.array(STRESS_ROOM_LO) * dim, np.array(STRESS_ROOM_HI) * dim
    with np.errstate(divide="ignore", invalid="ignore"):
        t_axis = np.where(d > 0, (hi - o) / d, np.where(d < 0, (lo - o) / d, np.inf))
    depth = t_axis.min(axis=-1)
    for c, r in STRESS_SPHERES:
        depth = np.minimum(depth, _ray_sphere(o, d, np.array(c) * dim, r * dim))
    depth = np.minimum(depth, _ray_box_outside(o, d, np.array(STRESS_PILLAR[0]) * dim, np.array(STRESS_PILLAR[1]) * dim))
    return depth


class NoiseStream:
    """sigma-scaled Gaussian samples from one mt19937(seed) stream: z = sqrt(-2 ln u1) cos(2 pi u2) with
    u1 = (w1 + 1) / 2^32, u2 = w2 / 2^32 from consecutive raw 32-bit words (float64)."""

    def __init__(self, seed: int = NOISE_SEED):
        self._bg = np.random.MT19937()
        self._bg._legacy_seeding(seed)

    def normal(self, n: int) -> np.ndarray:
        raw = self._bg.random_raw(2 * n).astype(np.float64).reshape(n, 2)
        u1 = (raw[:, 0] + 1.0) / 4294967296.0
        u2 = raw[:, 1] / 4294967296.0
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


class StressStream:
    """Same interface as SyntheticStream (``depth(f)`` in frame order, ``pose(f)``, ``k``); ICL-NUIM intrinsics with
    negative fy by default (BASELINE.json configs[0] / [2])."""

    def __init__(self, width: int, height: int, dim: float, holes: bool = True, noise: bool = True, negative_fy: bool = True, time_scale: float = 1.0, start: float = 0.0):
        self.width, self.height, self.dim = width, height, float(dim)
        self.negative_fy = negative_fy
        # time_scale < 1 samples the same camera path more densely (0.25: ~3 mm and 0.5 deg per frame, what a 30 Hz hand-held
        # sensor delivers -- the regime in which the reference's ICP tracks; at 1.0 it rejects every frame, tests/test_gpu_tracking.py)
        self.time_scale = float(time_scale)
        self.start = float(start)     # path position of frame 0 (in frames of the unscaled path)
        self.k = intrinsics(width, negative_fy)
        self._holes = HoleStream() if holes else None
        self._noise = NoiseStream() if noise else None
        self._next = 0

    def depth(self, frame: int) -> np.ndarray:
        if frame != self._next:
            raise ValueError("StressStream frames must be requested in order")
        self._next += 1
        mm = render_stress_depth(self._time(frame), self.width, self.height, self.dim, self.negative_fy) * 1000.0
        if self._noise is not None:
            mm = mm + NOISE_SIGMA_MM * self._noise.normal(self.width * self.height).reshape(self.height, self.width)
        mm = np.clip(np.floor(mm), 0, 65535).astype(np.uint16)
        d = mm.astype(np.float32) / np.float32(1000.0)
        if self._holes is not None:
            u = self._holes.uniform(self.width * self.height).reshape(self.height, self.width)
            d = np.where(u < HOLE_FRACTION, np.float32(0), d)
        return np.ascontiguousarray(d, dtype=np.float32)

    def pose(self, frame: int) -> np.ndarray:
        return stress_pose(self._time(frame), self.dim)

    def _time(self, frame: int):
        return frame if (self.time_scale == 1.0 and self.start == 0.0) else self.start + frame * self.time_scale


def stress_surface_distance(points: np.ndarray, dim: float) -> np.ndarray:
    """Distance of world points to the nearest analytic surface of the stress scene."""
    p = np.asarray(points, dtype=np.float64)
    lo, hi = np.array(STRESS_ROOM_LO) * dim, np.array(STRESS_ROOM_HI) * dim
    dist = np.minimum(np.abs(p - lo), np.abs(hi - p)).min(axis=-1)
    for c, r in STRESS_SPHERES:
        dist = np.minimum(dist, np.abs(np.linalg.norm(p - np.array(c) * dim, axis=-1) - r * dim))
    blo, bhi = np.array(STRESS_PILLAR[0]) * dim, np.array(STRESS_PILLAR[1]) * dim
    q = np.maximum(np.maximum(blo - p, p - bhi), 0.0)          # outside distance to the pillar box
    inside = np.minimum(p - blo, bhi - p).min(axis=-1)          # > 0 inside
    dist = np.minimum(dist, np.where(inside > 0, inside, np.linalg.norm(q, axis=-1)))
    return dist


def make_stream(kind: str, width: int, height: int, dim: float, **kw):
    """``kind``: "room" (SURVEY 8(d) box room + sphere) or "stress" (above)."""
    if kind == "stress":
        return StressStream(width, height, dim, **kw)
    if kind == "room":
        return SyntheticStream(width, height, dim, **kw)
    raise ValueError("unknown stream kind: " + kind)
