"""Host-side mirror of the reference's DenseSLAMSystem hot-path interface over the C ABI.

``DenseSLAMPipeline`` keeps the reference's method names and semantics for the path this
repository implements (se_denseslam/include/se/DenseSLAMSystem.h:193-212, 295, 353):
``integration(k, integration_rate, mu, frame)`` and ``raycasting(k, mu, frame)`` return the
reference's "did this stage run" booleans, ``setPose`` injects the camera pose, ``getMap``-style
read-back comes from ``blocks()`` / ``nodes()``.  All compute happens in libse_hip.so
(hand-written HIP for gfx950); there is no CPU fallback -- loading fails loudly without it.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

SDF, OFUSION = 0, 1
KERNELS = ("alloc_scan", "alloc_commit", "integrate", "raycast", "apply_bricks")
STAT_NAMES = ("probes", "new_keys", "swept", "nodes", "gets", "interps", "grads", "hits",
              "clk_iter", "clk_march", "clk_grad", "clk_wave_max", "clk_stage", "r13", "r14", "r15")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_LIB = None
# The per-frame entry points (se_hip_frame, se_hip_frame_tracked, se_hip_integrate, se_hip_raycast, se_hip_track, ...) take the pose, the
# intrinsics and the pyramid as plain addresses: an ndpointer argument costs ctypes 3 - 5 us of Python per call (type, dtype and flag checks),
# two of them were 10 % of a closed-loop frame.  DenseSLAMPipeline checks an array once (_addr) and remembers its address for as long as it
# holds the array.


class SeHipError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("volume_resolution", C.c_int32),
                ("volume_dimension", C.c_float), ("field_type", C.c_int32), ("device", C.c_int32),
                ("max_blocks", C.c_int64), ("row_begin", C.c_int32), ("row_end", C.c_int32)]


EXPORTS = {
    "se_hip_create": (C.c_int, [C.POINTER(_Config), C.POINTER(C.c_void_p)]),
    "se_hip_destroy": (C.c_int, [C.c_void_p]),
    "se_hip_last_error": (C.c_char_p, []),
    "se_hip_sync": (C.c_int, [C.c_void_p]),
    "se_hip_memory_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "se_hip_set_pinned_input": (C.c_int, [C.c_void_p, C.c_int32]),
    "se_hip_host_alloc": (C.c_void_p, [C.c_size_t]),
    "se_hip_host_free": (None, [C.c_void_p]),
    "se_hip_clear_overflow": (C.c_int, [C.c_void_p]),
    "se_hip_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "se_hip_set_scan_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "se_hip_scan_overlaps": (C.c_int, [C.c_void_p]),
    "se_hip_frame_is_fused": (C.c_int, [C.c_void_p]),
    "se_hip_set_streaming": (C.c_int, [C.c_void_p, C.c_int32]),
    "se_hip_set_image_ring": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "se_hip_raycast_deferred": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32]),
    "se_hip_get_launch_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    "se_hip_upload_depth": (C.c_int, [C.c_void_p, _f32p]),
    "se_hip_upload_depth_mm": (C.c_int, [C.c_void_p, np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS"), C.c_int32, C.c_int32]),
    "se_hip_set_depth_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "se_hip_integrate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_uint32]),
    "se_hip_alloc_scan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_uint32]),
    "se_hip_new_keys_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "se_hip_set_new_keys_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "se_hip_alloc_commit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64]),
    "se_hip_set_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "se_hip_alloc_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "se_hip_integrate_sweep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_uint32]),
    "se_hip_sweep_shard_bytes": (C.c_size_t, [C.c_size_t]),
    "se_hip_set_sweep_shard": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t]),
    "se_hip_apply_bricks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "se_hip_brick_exchange": (C.c_int, [C.c_void_p, C.c_void_p]),
    "se_hip_raycast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32]),
    "se_hip_image_tile_bytes": (C.c_size_t, [C.c_void_p, C.c_int32]),
    "se_hip_pack_image_tile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "se_hip_apply_image_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "se_hip_gather_images": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "se_hip_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_uint32]),
    "se_hip_download_vertex_normal": (C.c_int, [C.c_void_p, _f32p, _f32p]),
    "se_hip_vertex_normal_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "se_hip_frame_tracked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_float, C.c_uint32]),
    "se_hip_track": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p]),
    "se_hip_filter_depth": (C.c_int, [C.c_void_p, C.c_int32]),
    "se_hip_download_scaled_depth": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "se_hip_download_track": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "se_hip_render_volume": (C.c_int, [C.c_void_p, C.c_void_p, _f32p, _f32p, C.c_float, C.c_float, C.c_uint32, C.c_uint32]),
    "se_hip_render_depth": (C.c_int, [C.c_void_p, C.c_void_p]),
    "se_hip_render_track": (C.c_int, [C.c_void_p, C.c_void_p]),
    "se_hip_save_map": (C.c_int, [C.c_void_p, C.c_char_p]),
    "se_hip_load_map": (C.c_int, [C.c_void_p, C.c_char_p]),
    "se_hip_create_replicas": (C.c_int, [C.POINTER(_Config), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]),
    "se_hip_mesh_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "se_hip_mesh_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "se_hip_dump_mesh": (C.c_int, [C.c_void_p, C.c_char_p]),
    "se_hip_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "se_hip_download_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "se_hip_download_nodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "se_hip_enable_timing": (C.c_int, [C.c_void_p, C.c_int32]),
    "se_hip_get_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]),
    "se_hip_enable_stats": (C.c_int, [C.c_void_p, C.c_int32]),
    "se_hip_get_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int32]),
}


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so / libhsa-runtime64.so
    (torch/lib, found through an RPATH), libse_hip.so is linked against /opt/rocm's.  Loaded side by side, the copy that
    initialises second finds the GPU taken ("No HIP GPUs are available" from torch when libse_hip.so ran first).  Both
    copies carry the soname libamdhip64.so.7, so mapping torch's copy by path before libse_hip.so makes the dynamic
    loader resolve both users to that one object, whichever of them touches the GPU first.  No torch installed (or
    already imported: the soname is then mapped): nothing to do."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for base in (spec.submodule_search_locations or []) if spec else []:
        cand = os.path.join(base, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            return


def load_library(rebuild: bool = True):
    """Load libse_hip.so (building it with hipcc first if it is missing or stale)."""
    global _LIB
    if _LIB is None:
        _share_torch_hip_runtime()
        path = _build.LIB
        alt = os.environ.get("SE_HIP_LIB")   # A/B of kernel variants built to another path (tools/)
        if alt:
            import sys
            print(f"supereight_amd: SE_HIP_LIB is set -- loading {alt} instead of the in-tree libse_hip.so (A/B tooling)", file=sys.stderr)
            path, rebuild = alt, False
        if rebuild:
            try:
                path = _build.build()
            except RuntimeError:
                if not os.path.exists(path):
                    raise
        lib = C.CDLL(path)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
            fn.restype, fn.argtypes = res, args
        _LIB = lib
    return _LIB


def _colmajor(m) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(m, dtype=np.float32).reshape(4, 4).T).reshape(16)


class DenseSLAMPipeline:
    # (class-level defaults: an object that adopts a handle made by se_hip_create_replicas does not run __init__)
    _k_last = _k_arr = _pyr_arr = None
    _k_addr = 0

    def __init__(self, input_size, volume_resolution: int, volume_dimension: float, init_pose=None,
                 field_type: int = SDF, device: int = 0, max_blocks: int = 0, rows=None, streaming: bool = False):
        self.lib = load_library()
        self.W, self.H = int(input_size[0]), int(input_size[1])
        self.size, self.dim, self.field = int(volume_resolution), float(volume_dimension), field_type
        rb, re_ = rows if rows is not None else (0, 0)
        cfg = _Config(self.W, self.H, self.size, self.dim, field_type, device, max_blocks, rb, re_)
        h = C.c_void_p()
        self._h = None
        self._check(self.lib.se_hip_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.pose_ = np.eye(4, dtype=np.float32) if init_pose is None else init_pose
        self._keepalive = None
        self._ring_keepalive = None
        self._pinned = []
        if streaming:
            self.set_streaming(True)

    # pose_ (camera -> world, row-major 4x4) and its column-major copy for the C ABI; assign, do not
    # modify in place
    @property
    def pose_(self):
        if self._pose is None:      # the tracker updated the column-major copy in place (tracking, frame_tracked)
            self._pose = self._pose_cm.reshape(4, 4).T.copy()
        return self._pose

    @pose_.setter
    def pose_(self, m):
        self._pose = np.array(m, dtype=np.float32).reshape(4, 4)
        self._pose_cm = np.ascontiguousarray(self._pose.T).reshape(16)
        self._pose_cm_addr = self._pose_cm.ctypes.data

    @staticmethod
    def _addr(a, dtype, size: int) -> int:
        """Address of a C-contiguous array of `size` elements of `dtype` (checked here: the C ABI takes plain pointers)."""
        if not (type(a) is np.ndarray and a.dtype == dtype and a.size == size and a.flags.c_contiguous):
            raise TypeError(f"expected a C-contiguous {np.dtype(dtype).name}[{size}] array")
        return a.ctypes.data

    def _k(self, k) -> int:
        """Address of the float32[4] intrinsics (fx, fy, cx, cy) for the C ABI.  An int is taken as the address of such an array that the
        caller keeps alive (DenseSLAMPipeline.addr).  A float32[4] array is used in place and its address remembered while the caller
        keeps passing the same object; anything else is converted (and held) on every call."""
        if type(k) is int:
            return k
        if isinstance(k, np.integer):
            return int(k)
        if k is self._k_last:
            return self._k_addr
        if type(k) is np.ndarray and k.dtype == np.float32 and k.size == 4 and k.flags.c_contiguous:
            self._k_last, self._k_arr = k, k
        else:
            self._k_last, self._k_arr = None, np.ascontiguousarray(k, dtype=np.float32).reshape(4)
        self._k_addr = self._k_arr.ctypes.data
        return self._k_addr

    @staticmethod
    def addr(a) -> int:
        """Address of a float32 array for the entry points that accept plain addresses (frame, frame_tracked): check once, call many
        times.  The caller keeps the array alive and unchanged in size."""
        if not (type(a) is np.ndarray and a.dtype == np.float32 and a.flags.c_contiguous):
            raise TypeError("expected a C-contiguous float32 array")
        return a.ctypes.data

    # ------------------------------------------------------------------ plumbing
    def _check(self, status: int) -> int:
        if status < 0:
            raise SeHipError(f"se_hip error {status}: {self.lib.se_hip_last_error().decode()}")
        return status

    def close(self):
        if getattr(self, "_h", None):
            self.lib.se_hip_destroy(self._h)
            self._h = None
            for ptr in getattr(self, "_pinned", []):
                self.lib.se_hip_host_free(ptr)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self.lib.se_hip_sync(self._h))

    def clear_overflow(self) -> int:
        """Acknowledge a sticky SE_HIP_E_CAPACITY; returns the pending code (0 none, 1 pool, 2 key list, 3 brick segment)."""
        return self._check(self.lib.se_hip_clear_overflow(self._h))

    def set_stream(self, hip_stream_ptr: int):
        self._check(self.lib.se_hip_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def scan_overlaps(self) -> bool:
        return bool(self._check(self.lib.se_hip_scan_overlaps(self._h)))

    def frame_is_fused(self) -> bool:
        """True if frame() runs the one-queue streaming schedule (deferred raycast + next frame's scan in one launch, include/se_hip.h)."""
        return bool(self._check(self.lib.se_hip_frame_is_fused(self._h)))

    def set_streaming(self, on: bool = True) -> bool:
        """Opt into the one-queue streaming schedule (se_hip_set_streaming): frame() / raycasting_deferred() hold a frame's raycast back until the
        next frame's allocation scan.  Returns True if this handle will actually fuse."""
        return bool(self._check(self.lib.se_hip_set_streaming(self._h, int(on))))

    def set_image_ring(self, ptr: int, slots: int, keepalive=None):
        """vertex_ / normal_ of frame f go to slot f % slots of a caller-owned device ring (se_hip_set_image_ring); ptr = 0 restores the own images."""
        self._check(self.lib.se_hip_set_image_ring(self._h, C.c_void_p(ptr), slots))
        self._ring_keepalive = keepalive

    def launch_counts(self, reset: bool = False) -> dict:
        """Kernel launches per kind since the last reset, counted at enqueue time (does not flush a deferred raycast); 'pending' = a raycast is held back."""
        n = (C.c_int64 * (len(KERNELS) + 1))()
        pend = self._check(self.lib.se_hip_get_launch_counts(self._h, n, int(reset)))
        out = {k: int(n[i]) for i, k in enumerate(KERNELS)}
        out["fused"] = int(n[len(KERNELS)])
        out["pending"] = bool(pend)
        return out

    def set_scan_stream(self, hip_stream_ptr: int):
        self._check(self.lib.se_hip_set_scan_stream(self._h, C.c_void_p(hip_stream_ptr)))

    # ------------------------------------------------------------------ reference-shaped API
    def setPose(self, pose):
        """DenseSLAMSystem::setPose semantics minus the init-pose offset: pose is camera->world."""
        self.pose_ = pose

    def getPose(self):
        return self.pose_.copy()

    def set_depth(self, depth_m):
        """float_depth_ of the reference: host float32 metres (H, W)."""
        d = np.ascontiguousarray(depth_m, dtype=np.float32).reshape(-1)
        assert d.size == self.W * self.H
        self._check(self.lib.se_hip_upload_depth(self._h, d))

    def set_depth_mm(self, depth_mm):
        """preprocessing()'s mm2metersKernel fused into the upload: uint16 millimetres (h, w)."""
        d = np.ascontiguousarray(depth_mm, dtype=np.uint16)
        self._check(self.lib.se_hip_upload_depth_mm(self._h, d.reshape(-1), d.shape[1], d.shape[0]))

    def set_depth_device(self, ptr: int, keepalive=None):
        """Zero-copy: a float32 depth image already in HBM (e.g. a torch tensor's data_ptr())."""
        self._keepalive = keepalive
        self._check(self.lib.se_hip_set_depth_device(self._h, C.c_void_p(ptr)))

    def integration(self, k, integration_rate: int, mu: float, frame: int) -> bool:
        return bool(self._check(self.lib.se_hip_integrate(self._h, self._pose_cm_addr, self._k(k),
                                                          integration_rate, mu, frame)))

    def frame(self, depth_ptr: int, pose_cm, k, mu: float, frame: int, integration_rate: int = 1) -> int:
        """One frame in one FFI call: device depth pointer + integration() + raycasting().  pose_cm = the camera->world pose as
        16 float32 in column-major order (to_colmajor(pose)) or the address of such an array (DenseSLAMPipeline.addr: checked once by
        the caller, who keeps it alive -- the closed loop's per-frame Python cost is what the host adds to the frame); k likewise.
        Returns bit 0 = integrated, bit 1 = raycast."""
        if type(pose_cm) is not int:
            pose_cm = int(pose_cm) if isinstance(pose_cm, np.integer) else self._addr(pose_cm, np.float32, 16)
        return self._check(self.lib.se_hip_frame(self._h, depth_ptr, pose_cm, self._k(k), integration_rate, mu, frame))

    def raycasting(self, k, mu: float, frame: int) -> bool:
        return bool(self._check(self.lib.se_hip_raycast(self._h, self._pose_cm_addr, self._k(k), mu, frame)))

    def raycasting_deferred(self, k, mu: float, frame: int) -> bool:
        """raycasting() of a streaming caller: on a handle that fuses, launched together with the next integration()'s allocation scan."""
        return bool(self._check(self.lib.se_hip_raycast_deferred(self._h, self._pose_cm_addr, self._k(k), mu, frame)))

    def mesh(self) -> np.ndarray:
        """Marching-cubes triangles of the map, (n, 3, 3) float32 vertices in metres (order unspecified)."""
        n = C.c_int64()
        self._check(self.lib.se_hip_mesh_count(self._h, C.byref(n)))
        out = np.empty((n.value, 3, 3), np.float32)
        if n.value:
            w = C.c_int64()
            self._check(self.lib.se_hip_mesh_download(self._h, out.ctypes.data, n.value, C.byref(w)))
            out = out[: w.value]
        return out

    def dump_mesh(self, filename: str):
        """DenseSLAMSystem::dump_mesh: VTK polydata file."""
        self._check(self.lib.se_hip_dump_mesh(self._h, filename.encode()))

    def filter_depth(self, on: bool = True):
        """preprocessing(..., filterInput): tracking works on the bilateral-filtered depth image."""
        self._check(self.lib.se_hip_filter_depth(self._h, int(on)))

    def scaled_depth(self, level: int = 0) -> np.ndarray:
        out = np.empty((self.H >> level, self.W >> level), np.float32)
        self._check(self.lib.se_hip_download_scaled_depth(self._h, level, out.ctypes.data))
        return out

    TRACK_DTYPE = np.dtype([("result", np.int32), ("error", np.float32), ("J", np.float32, 6)])

    _PYRAMID = np.asarray((10, 5, 4), np.int32)
    _PYRAMID_ADDR = _PYRAMID.ctypes.data

    def _pyr(self, pyramid):
        if pyramid is None or pyramid is self._PYRAMID or tuple(pyramid) == (10, 5, 4):
            return self._PYRAMID_ADDR, 3
        self._pyr_arr = np.ascontiguousarray(pyramid, dtype=np.int32).reshape(-1)
        return self._pyr_arr.ctypes.data, self._pyr_arr.size

    def tracking(self, k, icp_threshold: float, tracking_rate: int, frame: int, pyramid=None) -> bool:
        """DenseSLAMSystem::tracking: ICP of the current depth image against the last raycast; updates pose_.
        pyramid = iterations per level, finest first (default (10, 5, 4))."""
        pa, n = self._pyr(pyramid)
        r = self._check(self.lib.se_hip_track(self._h, self._k(k), icp_threshold, tracking_rate, frame, pa, n, self._pose_cm_addr))
        self._pose = None      # (the column-major copy was updated in place; restored by the library if the check failed)
        return bool(r)

    def frame_tracked(self, depth_ptr: int, k, mu: float, frame: int, icp_threshold: float = 1e-5, tracking_rate: int = 1,
                      integration_rate: int = 1, pyramid=None) -> int:
        """One frame of the reference's loop with tracking on (se_apps/src/benchmark.cpp:115-150) in one FFI call: device depth
        pointer, tracked = tracking(); if tracked or frame <= 3: integration(); raycasting().  pose_ is updated.  Returns bit 0 =
        integrated, bit 1 = raycast, bit 2 = tracked."""
        pa, n = self._pyr(pyramid)
        r = self._check(self.lib.se_hip_frame_tracked(self._h, depth_ptr, self._k(k), icp_threshold, tracking_rate, pa, n,
                                                      self._pose_cm_addr, integration_rate, mu, frame))
        self._pose = None
        return r

    def track_data(self):
        t = np.zeros(self.W * self.H, self.TRACK_DTYPE)
        red = np.zeros(32, np.float32)
        it = C.c_int32()
        self._check(self.lib.se_hip_download_track(self._h, t.ctypes.data, red.ctypes.data, C.byref(it)))
        return t.reshape(self.H, self.W), red, it.value

    # the render*() methods (DenseSLAMSystem.h:241-286): RGBW uint8 images
    def renderVolume(self, view_pose, k, mu, largestep, frame=0, rate=1):
        out = np.zeros((self.H, self.W, 4), np.uint8)
        ran = self._check(self.lib.se_hip_render_volume(self._h, out.ctypes.data, _colmajor(view_pose), np.asarray(k, np.float32), mu, largestep, frame, rate))
        return out if ran else None

    def renderDepth(self):
        out = np.zeros((self.H, self.W, 4), np.uint8)
        self._check(self.lib.se_hip_render_depth(self._h, out.ctypes.data))
        return out

    def renderTrack(self):
        out = np.zeros((self.H, self.W, 4), np.uint8)
        self._check(self.lib.se_hip_render_track(self._h, out.ctypes.data))
        return out

    # stage split used by the multi-GPU driver
    def alloc_scan(self, k, integration_rate: int, mu: float, frame: int) -> bool:
        return bool(self._check(self.lib.se_hip_alloc_scan(self._h, self._pose_cm_addr, self._k(k),
                                                           integration_rate, mu, frame)))

    def new_keys_device(self):
        ptr, cap = C.c_void_p(), C.c_int64()
        self._check(self.lib.se_hip_new_keys_device(self._h, C.byref(ptr), C.byref(cap)))
        return ptr.value, cap.value

    def set_new_keys_buffer(self, ptr: int, capacity_words: int, keepalive=None):
        self._keys_keepalive = keepalive
        self._check(self.lib.se_hip_set_new_keys_buffer(self._h, C.c_void_p(ptr), capacity_words))

    def alloc_commit(self, lists_ptr: int, nlists: int, stride_words: int):
        self._check(self.lib.se_hip_alloc_commit(self._h, C.c_void_p(lists_ptr), nlists, stride_words))

    def set_exchange(self, nccl_comm: int, nccl_all_gather: int, world: int):
        self._check(self.lib.se_hip_set_exchange(self._h, C.c_void_p(nccl_comm), C.c_void_p(nccl_all_gather), world))

    def alloc_exchange(self, recv_ptr: int, words: int):
        self._check(self.lib.se_hip_alloc_exchange(self._h, C.c_void_p(recv_ptr), words))

    def sweep_shard_bytes(self, cap_bricks: int) -> int:
        return int(self.lib.se_hip_sweep_shard_bytes(cap_bricks))

    def set_sweep_shard(self, rank: int, world: int, send_ptr: int, cap_bricks: int, keepalive=None):
        """Sharded sweep (SURVEY 8e option 4): this replica integrates the blocks it owns and packs them into `send_ptr`."""
        self._check(self.lib.se_hip_set_sweep_shard(self._h, rank, world, C.c_void_p(send_ptr), cap_bricks))
        self._shard_keepalive = keepalive

    def apply_bricks(self, recv_ptr: int, world: int):
        self._check(self.lib.se_hip_apply_bricks(self._h, C.c_void_p(recv_ptr), world))

    def brick_exchange(self, recv_ptr: int):
        self._check(self.lib.se_hip_brick_exchange(self._h, C.c_void_p(recv_ptr)))

    def integrate_sweep(self, k, integration_rate: int, mu: float, frame: int) -> bool:
        return bool(self._check(self.lib.se_hip_integrate_sweep(self._h, self._pose_cm_addr, self._k(k),
                                                                integration_rate, mu, frame)))

    # ------------------------------------------------------------------ outputs
    def vertex_normal(self):
        v = np.zeros((self.H, self.W, 3), np.float32)
        n = np.zeros((self.H, self.W, 3), np.float32)
        self._check(self.lib.se_hip_download_vertex_normal(self._h, v.reshape(-1), n.reshape(-1)))
        return v, n

    def vertex_normal_device(self):
        v, n = C.c_void_p(), C.c_void_p()
        self._check(self.lib.se_hip_vertex_normal_device(self._h, C.byref(v), C.byref(n)))
        return v.value, n.value

    def set_pinned_input(self, on: bool = True):
        """Opt in to zero-copy host input: page-locked images handed to set_depth / set_depth_mm (pinned_image below) are read in place."""
        self._check(self.lib.se_hip_set_pinned_input(self._h, 1 if on else 0))

    def pinned_image(self, dtype=np.float32, shape=None):
        """A page-locked numpy image of the computation size (se_hip_host_alloc); freed with the pipeline."""
        shape = shape or (self.H, self.W)
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = self.lib.se_hip_host_alloc(nbytes)
        if not ptr:
            raise MemoryError("se_hip_host_alloc failed")
        self._pinned.append(ptr)
        return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)).view(dtype).reshape(shape)

    def memory_info(self) -> dict:
        """Layout and device memory of the map: dense brick grid or pooled bricks, brick slots, bytes of the bricks / of the whole replica."""
        out = (C.c_int64 * 4)()
        self._check(self.lib.se_hip_memory_info(self._h, out))
        return {"layout": "dense brick grid" if out[0] else "pooled bricks", "brick_slots": int(out[1]), "brick_bytes": int(out[2]), "device_bytes": int(out[3])}

    def counts(self):
        nb, nn = C.c_int32(), C.c_int32()
        self._check(self.lib.se_hip_counts(self._h, C.byref(nb), C.byref(nn)))
        return nb.value, nn.value

    def blocks(self):
        nb, _ = self.counts()
        coords = np.zeros((nb, 3), np.int32)
        x = np.zeros((nb, 512), np.float32)
        y = np.zeros((nb, 512), np.float32)
        act = np.zeros(nb, np.uint8)
        if nb:
            self._check(self.lib.se_hip_download_blocks(self._h, coords.ctypes.data, x.ctypes.data, y.ctypes.data, act.ctypes.data))
        return coords, x, y, act

    # ---- full vertex_ / normal_ images on every rank of a row-sharded run (include/se_hip.h)
    def image_tile_bytes(self, max_rows: int) -> int:
        return int(self.lib.se_hip_image_tile_bytes(self._h, max_rows))

    def pack_image_tile(self, send_ptr: int, max_rows: int):
        self._check(self.lib.se_hip_pack_image_tile(self._h, C.c_void_p(send_ptr), max_rows))

    def apply_image_tiles(self, recv_ptr: int, parts, max_rows: int):
        b = np.ascontiguousarray([q[0] for q in parts], np.int32)
        e = np.ascontiguousarray([q[1] for q in parts], np.int32)
        self._check(self.lib.se_hip_apply_image_tiles(self._h, C.c_void_p(recv_ptr), len(parts), max_rows, b.ctypes.data, e.ctypes.data))

    def gather_images(self, send_ptr: int, recv_ptr: int, parts, max_rows: int):
        b = np.ascontiguousarray([q[0] for q in parts], np.int32)
        e = np.ascontiguousarray([q[1] for q in parts], np.int32)
        self._check(self.lib.se_hip_gather_images(self._h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), max_rows, b.ctypes.data, e.ctypes.data))

    def block_flags(self):
        """coords[n,3] and VoxelBlock::active_[n] of the allocated blocks (sorted by key) without the voxel planes."""
        nb, _ = self.counts()
        coords = np.zeros((nb, 3), np.int32)
        act = np.zeros(nb, np.uint8)
        if nb:
            self._check(self.lib.se_hip_download_blocks(self._h, coords.ctypes.data, None, None, act.ctypes.data))
        return coords, act

    def nodes(self):
        _, nn = self.counts()
        code = np.zeros(nn, np.uint64)
        side = np.zeros(nn, np.uint32)
        x = np.zeros((nn, 8), np.float32)
        y = np.zeros((nn, 8), np.float32)
        self._check(self.lib.se_hip_download_nodes(self._h, code.ctypes.data, side.ctypes.data, x.ctypes.data, y.ctypes.data))
        return code, side, x, y

    def save(self, filename: str):
        """Octree::save of the reference (octree.hpp:898-914): same byte layout, entries sorted by key."""
        self._check(self.lib.se_hip_save_map(self._h, filename.encode()))

    def load(self, filename: str):
        """Octree::load counterpart (octree.hpp:917-950, minus its two defects): the map becomes what the file holds."""
        self._check(self.lib.se_hip_load_map(self._h, filename.encode()))

    # ------------------------------------------------------------------ measurement
    def enable_timing(self, on: bool = True):
        self._check(self.lib.se_hip_enable_timing(self._h, int(on)))

    def timings(self, reset: bool = False) -> dict:
        ms = (C.c_double * len(KERNELS))()
        n = (C.c_int64 * len(KERNELS))()
        self._check(self.lib.se_hip_get_timings(self._h, ms, n, int(reset)))
        return {k: {"ms_sum": ms[i], "launches": n[i]} for i, k in enumerate(KERNELS)}

    def enable_stats(self, on: bool = True):
        self._check(self.lib.se_hip_enable_stats(self._h, int(on)))

    def stats(self, reset: bool = False) -> dict:
        out = (C.c_uint64 * 16)()
        self._check(self.lib.se_hip_get_stats(self._h, out, int(reset)))
        return dict(zip(STAT_NAMES, (int(v) for v in out)))
