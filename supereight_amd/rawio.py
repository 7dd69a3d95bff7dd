"""SLAMBench ".raw" depth streams: the on-disk input format of the reference's apps.

Layout per frame (se_tools/scene2raw.cpp:170-176, reader se_apps/include/interface.h:384-426):
    uint32 w, h ; uint16 depth_mm[w*h] ; uint32 w, h ; uint8 rgb[w*h*3]
Frame stride = 16 + 2*w*h + 3*w*h bytes (1 536 016 at 640x480).  The depth frames go straight into
``DenseSLAMPipeline.set_depth_mm`` (mm2metersKernel fused into the upload)."""
from __future__ import annotations

import numpy as np


def write_raw(path: str, depth_mm_frames, rgb_frames=None) -> None:
    with open(path, "wb") as fh:
        for i, d in enumerate(depth_mm_frames):
            d = np.ascontiguousarray(d, dtype=np.uint16)
            h, w = d.shape
            np.array([w, h], np.uint32).tofile(fh)
            d.tofile(fh)
            np.array([w, h], np.uint32).tofile(fh)
            rgb = np.zeros((h, w, 3), np.uint8) if rgb_frames is None else np.ascontiguousarray(rgb_frames[i], np.uint8)
            rgb.tofile(fh)


def read_raw(path: str, with_rgb: bool = False):
    """Yields depth_mm (h, w) uint16 arrays (and rgb if asked) until the file ends."""
    with open(path, "rb") as fh:
        while True:
            hdr = np.fromfile(fh, np.uint32, 2)
            if hdr.size < 2:
                return
            w, h = int(hdr[0]), int(hdr[1])
            d = np.fromfile(fh, np.uint16, w * h)
            hdr2 = np.fromfile(fh, np.uint32, 2)
            rgb = np.fromfile(fh, np.uint8, w * h * 3)
            if d.size < w * h or hdr2.size < 2 or rgb.size < w * h * 3:
                return
            yield (d.reshape(h, w), rgb.reshape(h, w, 3)) if with_rgb else d.reshape(h, w)


def frame_stride(w: int, h: int) -> int:
    return 16 + 2 * w * h + 3 * w * h


def quaternion_to_rotation(w, x, y, z) -> np.ndarray:
    """Eigen::Quaternionf(w, x, y, z).toRotationMatrix() in float32 (no normalisation, as Eigen)."""
    f = np.float32
    w, x, y, z = f(w), f(x), f(y), f(z)
    tx, ty, tz = f(2) * x, f(2) * y, f(2) * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[f(1) - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, f(1) - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, f(1) - (txx + tyy)]], np.float32)


def read_groundtruth(path: str, init_pose=(0.0, 0.0, 0.0)):
    """Ground-truth trajectory as the reference consumes it: every non-comment line ends in
    ``tx ty tz qx qy qz qw`` (se_apps/include/interface.h:118-151, identity gt transform); the pose handed
    to the pipeline is ``setPose(gt)``: translation += init_pose (DenseSLAMSystem.h:353-356).
    Returns a list of camera->world 4x4 float32 matrices."""
    poses = []
    ip = np.asarray(init_pose, np.float32)
    with open(path) as fh:
        for line in fh:
            if not line.strip() or line[0] == "#":
                continue
            c = line.split()
            if len(c) < 7:
                raise ValueError("Invalid ground truth file format. Expected line format: ... tx ty tz qx qy qz qw")
            tx, ty, tz, qx, qy, qz, qw = (np.float32(v) for v in c[-7:])
            T = np.eye(4, dtype=np.float32)
            T[:3, :3] = quaternion_to_rotation(qw, qx, qy, qz)
            T[:3, 3] = np.array([tx, ty, tz], np.float32) + ip
            poses.append(T)
    return poses


def write_groundtruth(path: str, poses) -> None:
    """TUM-style ``timestamp tx ty tz qx qy qz qw`` lines from camera->world matrices (for tests)."""
    from scipy.spatial.transform import Rotation
    with open(path, "w") as fh:
        fh.write("# timestamp tx ty tz qx qy qz qw\n")
        for i, T in enumerate(poses):
            q = Rotation.from_matrix(np.asarray(T, np.float64)[:3, :3]).as_quat()   # x, y, z, w
            t = np.asarray(T, np.float64)[:3, 3]
            fh.write(f"{i} {t[0]:.9g} {t[1]:.9g} {t[2]:.9g} {q[0]:.9g} {q[1]:.9g} {q[2]:.9g} {q[3]:.9g}\n")


class RawStream:
    """Frame source over a SLAMBench .raw file + ground-truth trajectory with the interface of
    SyntheticStream (``depth(f)`` in metres, ``pose(f)``, ``k``): BASELINE.json configs 1 and 3."""

    def __init__(self, raw_path: str, traj_path: str, k, init_pose=(0.0, 0.0, 0.0), max_frames: int = 0):
        self.frames = []
        for d in read_raw(raw_path):
            self.frames.append(d)
            if max_frames and len(self.frames) >= max_frames:
                break
        if not self.frames:
            raise ValueError(f"{raw_path}: no frames")
        self.poses = read_groundtruth(traj_path, init_pose)
        if len(self.poses) < len(self.frames):
            raise ValueError(f"{traj_path}: {len(self.poses)} poses for {len(self.frames)} frames")
        self.height, self.width = self.frames[0].shape
        self.k = np.asarray(k, np.float32)

    def __len__(self):
        return len(self.frames)

    def depth(self, frame: int) -> np.ndarray:
        return self.frames[frame].astype(np.float32) / np.float32(1000.0)   # mm2metersKernel: depth / 1000.0f

    def pose(self, frame: int) -> np.ndarray:
        return self.poses[frame]


# ---------------------------------------------------------------------------------------------------------
# ICL-NUIM scene directories -> .raw : the converter the reference ships as se_tools/scene2raw.cpp (its README's
# data-set recipe, README.md:60-80).  ICL-NUIM `scene_00_NNNN.depth` files hold, per pixel and in row-major order, the
# EUCLIDEAN distance along the ray in metres (text, whitespace separated); the .raw stream holds z-depth in uint16
# millimetres.  Arithmetic as scene2raw.cpp:97-108: double precision, z = 1000 d / sqrt(((u-u0)/fx)^2 + ((v-v0)/fy)^2 + 1),
# truncated to uint16 by the store (the camera constants are float there: scene2raw.cpp:25-38).
ICL_SCENE_K = (np.float32(481.20), np.float32(-480.00), np.float32(319.50), np.float32(239.50))   # fx, fy, u0, v0 of the scene files


def icl_ray_length_to_depth_mm(dist_m, k=ICL_SCENE_K) -> np.ndarray:
    """(h, w) ray lengths in metres -> (h, w) uint16 z-depth in millimetres, as readDepthFile() stores them."""
    d = np.asarray(dist_m, np.float64) * 1000.0
    h, w = d.shape
    fx, fy, u0, v0 = (np.float32(v) for v in k)
    u = ((np.arange(w, dtype=np.float32) - u0) / fx).astype(np.float64)     # (u - _u0) / _focal_x is float arithmetic, stored in a double
    v = ((np.arange(h, dtype=np.float32) - v0) / fy).astype(np.float64)
    z = d / np.sqrt(u[None, :] * u[None, :] + v[:, None] * v[:, None] + 1.0)
    return z.astype(np.int64).astype(np.uint16)          # double -> ushort: truncation, modulo 2^16 like the C cast on x86-64


def read_icl_depth_file(path: str, width: int = 640, height: int = 480) -> np.ndarray:
    vals = np.loadtxt(path, dtype=np.float64).reshape(-1)
    if vals.size < width * height:
        raise ValueError(f"{path}: {vals.size} values, {width * height} expected")
    return vals[: width * height].reshape(height, width)   # (the reference also ignores one trailing value, scene2raw.cpp:91-92)


def scene2raw(scene_dir: str, out_path: str, width: int = 640, height: int = 480) -> int:
    """scene_00_0000.depth, scene_00_0001.depth, ... -> SLAMBench .raw (RGB planes zero: this path never reads them).
    Returns the number of frames written."""
    import os
    frames = []
    i = 0
    while True:
        f = os.path.join(scene_dir, f"scene_00_{i:04d}.depth")
        if not os.path.exists(f):
            break
        frames.append(icl_ray_length_to_depth_mm(read_icl_depth_file(f, width, height)))
        i += 1
    write_raw(out_path, frames)
    return i
