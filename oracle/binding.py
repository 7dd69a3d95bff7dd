"""ctypes binding of the CPU oracle (oracle/se_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under supereight_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS: dict = {}

c_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
c_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
c_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
c_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
c_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
c_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(native: bool = False, force: bool = False, fma: bool = False) -> str:
    """(Re)build the oracle shared library with oracle/Makefile; returns its path.
    fma=True: the noise-floor variant compiled with -ffp-contract=fast (see oracle/Makefile)."""
    out = "libse_oracle_fma.so" if fma else ("libse_oracle_native.so" if native else "libse_oracle.so")
    path = os.path.join(_HERE, out)
    src = os.path.join(_HERE, "se_oracle.cpp")
    stale = (not os.path.exists(path)) or os.path.getmtime(path) < os.path.getmtime(src)
    if force or stale:
        args = ["make", "-C", _HERE, f"OUT={out}"] + (["FPC=fast"] if fma else (["ARCH=native"] if native else []))
        if force:
            args.insert(1, "-B")
        subprocess.run(args, check=True, capture_output=True)
    return path


def _declare(lib):
    u64, i32, f32, vp = C.c_uint64, C.c_int, C.c_float, C.c_void_p
    sig = {
        "so_compute_morton": (u64, [u64, u64, u64]),
        "so_unpack_morton": (None, [u64, c_i32p]),
        "so_mask": (u64, [i32]),
        "so_encode": (u64, [i32, i32, i32, i32, i32]),
        "so_decode": (None, [u64, c_i32p]),
        "so_parent": (u64, [u64, i32]),
        "so_child_id": (i32, [u64, i32, i32]),
        "so_descendant": (i32, [u64, u64, i32]),
        "so_far_corner": (None, [u64, i32, i32, c_i32p]),
        "so_face_neighbour": (None, [u64, C.c_uint, C.c_uint, C.c_uint, c_i32p]),
        "so_exterior_neighbours": (None, [c_u64p, u64, i32, i32]),
        "so_siblings": (None, [c_u64p, u64, i32]),
        "so_unique": (i32, [c_u64p, i32]),
        "so_unique_u32": (i32, [c_u32p, i32]),
        "so_filter_ancestors": (i32, [c_u64p, i32, i32]),
        "so_unique_multiscale": (i32, [c_u64p, i32, C.c_uint]),
        "so_filter_ancestors_u32": (i32, [c_u32p, i32, i32]),
        "so_unique_multiscale_u32": (i32, [c_u32p, i32, C.c_uint]),
        "so_cvt_i32": (i32, [f32]),
        "so_bspline_lookup": (None, [c_f32p]),
        "so_hnew": (f32, [f32]),
        "so_update_logs": (f32, [f32, f32]),
        "so_ft_create": (vp, [i32, f32, f32, f32]),
        "so_ft_destroy": (None, [vp]),
        "so_ft_hash": (u64, [vp, i32, i32, i32, i32]),
        "so_ft_allocate": (i32, [vp, c_u64p, i32]),
        "so_ft_get": (f32, [vp, i32, i32, i32]),
        "so_ft_get_fine": (f32, [vp, i32, i32, i32]),
        "so_ft_set": (None, [vp, i32, i32, i32, f32]),
        "so_ft_fetch": (i32, [vp, i32, i32, i32, C.POINTER(u64), c_i32p]),
        "so_ft_fetch_octant": (i32, [vp, i32, i32, i32, i32, C.POINTER(u64), C.POINTER(C.c_uint), C.POINTER(i32)]),
        "so_ft_set_octant_value": (i32, [vp, i32, i32, i32, i32, i32, f32]),
        "so_ft_insert": (i32, [vp, i32, i32, i32, i32, C.POINTER(u64), c_i32p, C.POINTER(i32)]),
        "so_ft_counts": (None, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "so_ft_check_children_mask": (i32, [vp]),
        "so_ft_node_sides": (None, [vp, c_u32p, c_u64p]),
        "so_ft_gather": (None, [vp, i32, i32, i32, c_f32p]),
        "so_ft_interp": (f32, [vp, f32, f32, f32]),
        "so_ft_grad": (None, [vp, f32, f32, f32, c_f32p]),
        "so_ft_ray_blocks": (i32, [vp, c_f32p, c_f32p, f32, f32, c_u64p, c_f32p, c_f32p, i32, c_f32p]),
        "so_pipe_create": (vp, [i32, i32, f32, i32, i32]),
        "so_pipe_destroy": (None, [vp]),
        "so_pipe_count_stats": (None, [vp, i32]),
        "so_pipe_integrate": (i32, [vp, c_f32p, c_f32p, c_f32p, C.c_uint, f32, C.c_uint]),
        "so_pipe_raycast": (i32, [vp, c_f32p, c_f32p, f32, C.c_uint, c_f32p, c_f32p]),
        "so_pipe_scan": (C.c_uint, [vp, c_f32p, c_f32p, c_f32p, f32, C.c_uint]),
        "so_pipe_get_keys": (None, [vp, c_u64p, C.c_uint]),
        "so_pipe_allocate_keys": (None, [vp, c_u64p, C.c_uint]),
        "so_pipe_activate": (i32, [vp, c_i32p, i32]),
        "so_pipe_sweep": (None, [vp, c_f32p, c_f32p, c_f32p, f32, C.c_uint]),
        "so_pipe_counts": (None, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "so_pipe_get_blocks": (None, [vp, c_i32p, c_f32p, c_f32p, c_u8p]),
        "so_pipe_get_nodes": (None, [vp, c_u64p, c_u32p, c_f32p, c_f32p]),
        "so_pipe_stats": (None, [vp, c_u64p]),
        "so_pipe_timings": (None, [vp, c_f64p]),
        "so_bilateral_filter": (None, [c_f32p, c_f32p, i32, i32]),
        "so_half_sample": (None, [c_f32p, i32, i32, c_f32p, i32, f32, i32]),
        "so_depth2vertex": (None, [c_f32p, c_f32p, i32, i32, c_f32p]),
        "so_vertex2normal": (None, [c_f32p, c_f32p, i32, i32, i32]),
        "so_se3_exp": (None, [c_f32p, c_f32p]),
        "so_solve6": (i32, [c_f32p, c_f32p]),
        "so_tracking": (i32, [c_f32p, i32, i32, c_f32p, c_i32p, i32, f32, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, c_f32p, C.POINTER(i32)]),
        "so_render_depth": (None, [c_u8p, c_f32p, i32, i32]),
        "so_render_track": (None, [c_u8p, C.c_void_p, i32, i32]),
        "so_pipe_render_volume": (None, [vp, c_u8p, c_f32p, c_f32p, c_f32p, f32, f32, c_f32p, c_f32p]),
        "so_pipe_save": (i32, [vp, C.c_char_p]),
        "so_pipe_mesh": (C.c_longlong, [vp, C.c_void_p, C.c_longlong]),
        "so_num_threads": (i32, []),
        "so_set_num_threads": (None, [i32]),
        "so_set_sophus_quat": (None, [i32]),
        "so_set_libm_sincos": (None, [i32]),
        "so_fp_contract": (i32, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def load(native: bool = False, fma: bool = False):
    key = "fma" if fma else ("native" if native else "portable")
    if key not in _LIBS:
        try:
            path = build(native=native, fma=fma)
            lib = C.CDLL(path)
        except (OSError, subprocess.CalledProcessError):
            path = build(native=native, force=True, fma=fma)
            lib = C.CDLL(path)
        _LIBS[key] = _declare(lib)
    return _LIBS[key]


SDF, OFUSION = 0, 1
STAT_NAMES = ("probes", "keys_emitted", "swept", "nodes", "gets", "interps", "grads", "hits", "oob", "truncated", "oob_ub")


class OraclePipeline:
    """The reference's DenseSLAMSystem::integration / ::raycasting on the CPU oracle."""

    def __init__(self, field: int, size: int, dim: float, width: int, height: int, native: bool = False, fma: bool = False):
        self.lib = load(native, fma)
        self.field, self.size, self.dim, self.W, self.H = field, size, float(dim), width, height
        self.h = self.lib.so_pipe_create(field, size, dim, width, height)

    def close(self):
        if self.h:
            self.lib.so_pipe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def count_stats(self, on: bool = True):
        self.lib.so_pipe_count_stats(self.h, int(on))

    def integrate(self, depth, pose, k, mu, frame, rate=1) -> bool:
        from supereight_amd.synthetic import to_colmajor
        d = np.ascontiguousarray(depth, dtype=np.float32).reshape(-1)
        return bool(self.lib.so_pipe_integrate(self.h, d, to_colmajor(pose), np.asarray(k, np.float32), rate, mu, frame))

    # staged variants (multi-rank protocol tests)
    def scan_keys(self, depth, pose, k, mu, frame) -> np.ndarray:
        from supereight_amd.synthetic import to_colmajor
        d = np.ascontiguousarray(depth, dtype=np.float32).reshape(-1)
        n = self.lib.so_pipe_scan(self.h, d, to_colmajor(pose), np.asarray(k, np.float32), mu, frame)
        keys = np.zeros(n, np.uint64)
        if n:
            self.lib.so_pipe_get_keys(self.h, keys, n)
        return keys

    def allocate_keys(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        self.lib.so_pipe_allocate_keys(self.h, keys, len(keys))

    def activate(self, coords) -> int:
        c = np.ascontiguousarray(coords, dtype=np.int32).reshape(-1, 3)
        return self.lib.so_pipe_activate(self.h, c.reshape(-1), len(c)) if len(c) else 0

    def sweep(self, depth, pose, k, mu, frame):
        from supereight_amd.synthetic import to_colmajor
        d = np.ascontiguousarray(depth, dtype=np.float32).reshape(-1)
        self.lib.so_pipe_sweep(self.h, d, to_colmajor(pose), np.asarray(k, np.float32), mu, frame)

    def raycast(self, pose, k, mu, frame):
        from supereight_amd.synthetic import to_colmajor
        v = np.zeros((self.H, self.W, 3), np.float32)
        n = np.zeros((self.H, self.W, 3), np.float32)
        ran = bool(self.lib.so_pipe_raycast(self.h, to_colmajor(pose), np.asarray(k, np.float32), mu, frame,
                                            v.reshape(-1), n.reshape(-1)))
        return ran, v, n

    def render_volume(self, view_pose, raycast_pose, k, mu, largestep, vertex, normal):
        from supereight_amd.synthetic import to_colmajor
        out = np.zeros((self.H, self.W, 4), np.uint8)
        self.lib.so_pipe_render_volume(self.h, out.reshape(-1), to_colmajor(view_pose), to_colmajor(raycast_pose), np.asarray(k, np.float32),
                                       mu, largestep, np.ascontiguousarray(vertex, np.float32).reshape(-1),
                                       np.ascontiguousarray(normal, np.float32).reshape(-1))
        return out

    def save(self, filename: str) -> bool:
        return bool(self.lib.so_pipe_save(self.h, filename.encode()))

    def mesh(self) -> np.ndarray:
        """DenseSLAMSystem::dump_mesh up to the triangle list: (n, 3, 3) float32 vertices, block-pool order."""
        n = self.lib.so_pipe_mesh(self.h, None, 0)
        out = np.empty((n, 3, 3), np.float32)
        if n:
            self.lib.so_pipe_mesh(self.h, out.ctypes.data, n)
        return out

    def counts(self):
        nb, nn = C.c_int(), C.c_int()
        self.lib.so_pipe_counts(self.h, C.byref(nb), C.byref(nn))
        return nb.value, nn.value

    def blocks(self):
        """Blocks sorted by Morton key: coords[n,3], x[n,512], y[n,512], active[n]."""
        nb, _ = self.counts()
        coords = np.zeros((nb, 3), np.int32)
        x = np.zeros((nb, 512), np.float32)
        y = np.zeros((nb, 512), np.float32)
        act = np.zeros(nb, np.uint8)
        if nb:
            self.lib.so_pipe_get_blocks(self.h, coords.reshape(-1), x.reshape(-1), y.reshape(-1), act)
        return coords, x, y, act

    def nodes(self):
        _, nn = self.counts()
        code = np.zeros(nn, np.uint64)
        side = np.zeros(nn, np.uint32)
        x = np.zeros((nn, 8), np.float32)
        y = np.zeros((nn, 8), np.float32)
        self.lib.so_pipe_get_nodes(self.h, code, side, x.reshape(-1), y.reshape(-1))
        return code, side, x, y

    def stats(self) -> dict:
        out = np.zeros(11, np.uint64)
        self.lib.so_pipe_stats(self.h, out)
        return dict(zip(STAT_NAMES, (int(v) for v in out)))

    def timings(self) -> dict:
        out = np.zeros(4, np.float64)
        self.lib.so_pipe_timings(self.h, out)
        return dict(zip(("alloc_scan", "allocate", "sweep", "raycast"), out.tolist()))


TRACK_DTYPE = np.dtype([("result", np.int32), ("error", np.float32), ("J", np.float32, 6)])


def oracle_bilateral_filter(depth):
    """bilateralFilterKernel of preprocessing(..., filterInput=true): scaled_depth_[0] from float_depth_."""
    lib = load()
    H, W = depth.shape
    out = np.empty((H, W), np.float32)
    lib.so_bilateral_filter(out.reshape(-1), np.ascontiguousarray(depth, np.float32).reshape(-1), W, H)
    return out


def oracle_half_sample(depth, e_d: float, r: int = 1):
    """halfSampleRobustImageKernel (preprocessing.cpp:190-226): one pyramid level down."""
    lib = load()
    H, W = depth.shape
    out = np.empty((H // 2, W // 2), np.float32)
    lib.so_half_sample(out.reshape(-1), W // 2, H // 2, np.ascontiguousarray(depth, np.float32).reshape(-1), W, e_d, r)
    return out


def oracle_tracking(depth, k, pose, raycast_pose, ref_vertex, ref_normal, icp_threshold=1e-5, pyramid=(10, 5, 4), fma=False):
    """DenseSLAMSystem::tracking on the oracle.  Returns (tracked, new_pose 4x4, TrackData image, reduce row, iterations).
    fma=True: the noise-floor build (-ffp-contract=fast), see oracle/Makefile."""
    from supereight_amd.synthetic import to_colmajor
    lib = load(fma=fma)
    H, W = depth.shape
    pose_cm = to_colmajor(pose).copy()
    track = np.zeros(W * H, TRACK_DTYPE)
    red = np.zeros(32, np.float32)
    it = C.c_int()
    # the ICP is thousands of small parallel regions: on a 256-thread box the OpenMP barriers cost more than the work
    nthr = lib.so_num_threads()
    lib.so_set_num_threads(min(nthr, 16))
    ok = lib.so_tracking(np.ascontiguousarray(depth, np.float32).reshape(-1), W, H, np.asarray(k, np.float32),
                         np.asarray(pyramid, np.int32), len(pyramid), icp_threshold,
                         np.ascontiguousarray(ref_vertex, np.float32).reshape(-1), np.ascontiguousarray(ref_normal, np.float32).reshape(-1),
                         to_colmajor(raycast_pose), pose_cm, track.ctypes.data, red, C.byref(it))
    lib.so_set_num_threads(nthr)
    return bool(ok), pose_cm.reshape(4, 4).T.copy(), track.reshape(H, W), red, it.value
