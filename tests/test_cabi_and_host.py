"""CPU-only checks of the drop-in boundary: the shared library loads, exports every entry point that
include/se_hip.h declares, fails loudly without a GPU, and the host-side helpers behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "se_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(se_hip_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from supereight_amd import build
    lib = C.CDLL(build.build())
    names = declared_functions()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), f"libse_hip.so does not export {n}"


def test_binding_covers_the_header():
    from supereight_amd.pipeline import EXPORTS
    assert sorted(EXPORTS) == declared_functions()


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly (no silent CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from supereight_amd.pipeline import DenseSLAMPipeline, SeHipError
    with pytest.raises(SeHipError, match="no HIP device|no CPU path"):
        DenseSLAMPipeline((64, 48), 128, 1.2)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "supereight_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle|#include\s+\"[^\"]*oracle", text, flags=re.M), f"{f} uses the oracle"


def test_create_argument_validation():
    from supereight_amd.pipeline import _Config, load_library
    lib = load_library()
    h = C.c_void_p()
    for bad in (dict(width=0), dict(volume_resolution=500), dict(volume_resolution=32), dict(volume_dimension=0.0),
                dict(field_type=7)):
        kw = dict(width=64, height=48, volume_resolution=128, volume_dimension=1.2, field_type=0, device=0,
                  max_blocks=0, row_begin=0, row_end=0)
        kw.update(bad)
        cfg = _Config(**kw)
        assert lib.se_hip_create(C.byref(cfg), C.byref(h)) == -1   # SE_HIP_E_INVALID, before any device call
        assert lib.se_hip_last_error()
    assert lib.se_hip_create(None, C.byref(h)) == -1


def test_row_partition():
    from supereight_amd.multi_gpu import row_partition
    for H in (480, 960, 120, 8, 100):
        for R in (1, 2, 3, 4, 8):
            parts = row_partition(H, R)
            assert parts[0][0] == 0 and parts[-1][1] == H
            for (a, b), (c, d) in zip(parts, parts[1:]):
                assert b == c and a <= b
            for a, b in parts[:-1]:
                assert a % 8 == 0 and b % 8 == 0
    sizes = [b - a for a, b in row_partition(480, 8)]
    assert max(sizes) - min(sizes) <= 8


def test_synthetic_stream_properties():
    from supereight_amd.synthetic import HoleStream, SyntheticStream, intrinsics, pose, surface_distance, to_colmajor
    k = intrinsics(640)
    assert np.allclose(k, [481.2, 480, 320, 240])
    s = SyntheticStream(640, 480, 4.8)
    d0 = s.depth(0)
    assert d0.shape == (480, 640) and d0.dtype == np.float32
    holes = (d0 == 0).sum()
    assert holes == 6125                       # libstdc++ mt19937(54321) + uniform_real<float> < 0.02 (checked against g++)
    assert 1.0 < d0[d0 > 0].min() and d0.max() < 4.7
    with pytest.raises(ValueError):
        s.depth(5)
    P = pose(10, 4.8)
    assert np.allclose(P[:3, :3] @ P[:3, :3].T, np.eye(3), atol=1e-6)
    assert np.allclose(to_colmajor(P).reshape(4, 4).T, P)
    u = HoleStream().uniform(5)
    assert np.allclose(u, [0.911640763, 0.509720981, 0.623824835, 0.183354408, 0.791803837], atol=1e-8)
    pts = np.array([[0.05 * 4.8, 1.0, 1.0], [2.4, 2.4, 0.62 * 4.8 - 0.08 * 4.8]])
    assert np.allclose(surface_distance(pts, 4.8), 0, atol=1e-6)


def test_raw_stream_with_groundtruth_roundtrip(tmp_path):
    """BASELINE.json configs 1 / 3 harness: a SLAMBench .raw file + TUM-style ground truth read back as
    the frame source of bench.py (RawStream) reproduces the frames exactly and the poses to float accuracy;
    readNextPose's quaternion -> matrix and setPose's init-pose offset are the reference's
    (se_apps/include/interface.h:118-151, DenseSLAMSystem.h:353-356)."""
    from supereight_amd import rawio
    from supereight_amd.synthetic import SyntheticStream, render_depth_mm, intrinsics
    W, H, dim, F = 64, 48, 4.8, 5
    s = SyntheticStream(W, H, dim, holes=False, negative_fy=True)
    mm = [render_depth_mm(f, W, H, dim, negative_fy=True) for f in range(F)]
    init = np.array([0.34, 0.5, 0.24], np.float32) * np.float32(dim)
    rel = []
    for f in range(F):
        T = s.pose(f).copy(); T[:3, 3] -= init; rel.append(T)          # what a gt file holds: poses relative to the start
    rawio.write_raw(str(tmp_path / "s.raw"), mm)
    rawio.write_groundtruth(str(tmp_path / "s.gt"), rel)
    rs = rawio.RawStream(str(tmp_path / "s.raw"), str(tmp_path / "s.gt"), intrinsics(W, True), init)
    assert len(rs) == F and (rs.width, rs.height) == (W, H) and rs.k[1] < 0
    for f in range(F):
        assert (rs.depth(f) == s.depth(f)).all()
        assert np.abs(rs.pose(f) - s.pose(f)).max() < 2e-6
    # Eigen's toRotationMatrix for a known quaternion: 90 degrees about +z
    R = rawio.quaternion_to_rotation(np.sqrt(0.5), 0.0, 0.0, np.sqrt(0.5))
    assert np.abs(R - np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], np.float32)).max() < 1e-6
    with pytest.raises(ValueError):
        (tmp_path / "bad.gt").write_text("0 1 2 3\n")
        rawio.read_groundtruth(str(tmp_path / "bad.gt"))


def test_one_hip_runtime_per_process():
    """libse_hip.so first, torch second (the order that used to leave torch with "No HIP GPUs are available"): the
    process must end up with ONE libamdhip64 mapped (pipeline._share_torch_hip_runtime)."""
    import subprocess
    import sys
    code = ("from supereight_amd.pipeline import load_library\n"
            "load_library(rebuild=False)\n"
            "import torch\n"
            "m = open('/proc/self/maps').read().split('\\n')\n"
            "print(len({l.split()[-1] for l in m if 'libamdhip64' in l}), len({l.split()[-1] for l in m if 'libhsa-runtime64' in l}))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[-2:] == ["1", "1"], out.stdout


def test_pipeline_passes_remembered_addresses():
    """The per-frame entry points take pose / intrinsics / pyramid as plain addresses (an ndpointer argument costs ctypes microseconds of checks
    per call): DenseSLAMPipeline checks an array once, holds it and remembers where it lives.  Host logic only -- no library call."""
    import ctypes as C
    from supereight_amd.pipeline import EXPORTS, DenseSLAMPipeline
    p = DenseSLAMPipeline.__new__(DenseSLAMPipeline)          # (no handle: the helpers under test make no C call)
    # pose_: row-major 4x4 in, column-major copy held, its address remembered, the 4x4 rebuilt lazily after an in-place update by the library
    m = np.arange(16, dtype=np.float64).reshape(4, 4)
    p.pose_ = m
    assert p._pose_cm.dtype == np.float32 and p._pose_cm_addr == p._pose_cm.ctypes.data
    assert (p._pose_cm == m.T.reshape(16)).all() and (p.pose_ == m).all()
    p._pose_cm[12] = 42.0; p._pose = None                      # what tracking() / frame_tracked() do after the call
    assert p.pose_[0, 3] == 42.0 and p.getPose() is not p.pose_
    # intrinsics: a float32[4] array is used in place and remembered by identity; anything else is converted and held
    k = np.array([481.2, 480.0, 320.0, 240.0], np.float32)
    a = p._k(k)
    assert a == k.ctypes.data and p._k(k) == a
    k[0] = 500.0                                               # in-place change of the caller's array: the address still shows it
    assert C.cast(p._k(k), C.POINTER(C.c_float))[0] == 500.0
    b = p._k([1.0, 2.0, 3.0, 4.0])
    assert b != a and [C.cast(b, C.POINTER(C.c_float))[i] for i in range(4)] == [1.0, 2.0, 3.0, 4.0]
    assert p._k(k.astype(np.float64)) != a and p._k(k) == a and p._k(12345) == 12345 and p._k(np.int64(12345)) == 12345
    # pyramid: the default is one shared array; others are converted and held
    assert p._pyr(None) == p._pyr((10, 5, 4)) == (DenseSLAMPipeline._PYRAMID_ADDR, 3)
    pa, n = p._pyr([4, 3])
    assert n == 2 and [C.cast(pa, C.POINTER(C.c_int32))[i] for i in range(2)] == [4, 3]
    # addr(): what callers of frame() / frame_tracked() pass for arrays they hold themselves
    pose_cm = np.zeros(16, np.float32)
    assert DenseSLAMPipeline.addr(pose_cm) == pose_cm.ctypes.data
    with pytest.raises(TypeError):
        DenseSLAMPipeline.addr(np.zeros(16, np.float64))
    with pytest.raises(TypeError):
        DenseSLAMPipeline._addr(np.zeros(12, np.float32), np.float32, 16)
    # the declared argument types of the per-frame entry points are addresses, not array checkers
    for name in ("se_hip_frame", "se_hip_frame_tracked", "se_hip_integrate", "se_hip_raycast", "se_hip_track", "se_hip_alloc_scan", "se_hip_integrate_sweep"):
        assert all(t in (C.c_void_p, C.c_float, C.c_uint32, C.c_int32) for t in EXPORTS[name][1]), name
