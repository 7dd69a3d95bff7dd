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
