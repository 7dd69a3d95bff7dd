"""The sweep's shared-reciprocal divisions and its range-restricted square root (se_rcp_refined / se_div_refined / se_inv_refined /
se_sqrt_ge1, supereight_amd/csrc/se_kernels.h) against the compiler's own `/` and sqrtf, exhaustively on the GPU (tests/cpp/arith_check.hip):
  * x / z, 1 / z: all 2^23 numerator mantissas x both signs x the binades a volume can produce, for 600 divisors spread over [1e-4, 2^40]
    (edge mantissas included) -- bit-identical quotients wherever the IEEE sequence does not rescale, identical 1 + q^2 everywhere;
  * fminf(1, diff / mu): EVERY float diff the sweep can form, for the benchmark's mu values and the ends of the admitted range;
  * sqrtf(s): every float s >= 1, +inf, NaN.
They are the same machine operations minus steps that are the identity on these operands, so this is a check of the range argument,
not of a numerical approximation.  And end to end: the map after a stream swept with k_integrate<FAST = true> equals, bit for bit, the map
swept with the compiler's divisions (SE_HIP_IEEE_SWEEP=1) -- both already equal the oracle's in tests/test_gpu_parity.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "cpp", "libse_arith_check.so")
SRC = os.path.join(ROOT, "tests", "cpp", "arith_check.hip")


def build_arith_check(force: bool = False) -> str:
    from supereight_amd import build as hb
    deps = [SRC, os.path.join(hb.SRC_DIR, "se_kernels.h"), os.path.join(hb.SRC_DIR, "se_device.h")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.run([hb.hipcc()] + hb.FLAGS + ["-o", SO, SRC], check=True, capture_output=True)
    return SO


@pytest.fixture(scope="module")
def chk():
    from supereight_amd.pipeline import load_library
    load_library()      # (one HIP runtime per process: see pipeline._share_torch_hip_runtime)
    lib = C.CDLL(build_arith_check())
    lib.se_arith_check.restype = C.c_int
    lib.se_arith_check.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]

    def run(which, dens=(), exps=()):
        d = np.ascontiguousarray(dens, np.float32)
        e = np.ascontiguousarray(exps, np.int32)
        out = (C.c_uint64 * 5)()
        rc = lib.se_arith_check(which, d.ctypes.data if d.size else None, d.size, e.ctypes.data if e.size else None, e.size, out)
        assert rc == 0, rc
        as_f = lambda u: float(np.uint32(u).view(np.float32))
        return int(out[0]), {"a": hex(out[1]), "b": hex(out[2]), "got": hex(out[3]), "want": hex(out[4]), "a_f": as_f(out[1]), "b_f": as_f(out[2])}
    return run


def test_divisions_by_z(chk):
    rng = np.random.default_rng(7)
    z = np.exp2(rng.uniform(np.log2(1e-4), 40.0, 560)).astype(np.float32)
    edge = []
    for e in (-13, -3, 0, 1, 7, 39):      # 1.0, 1 + ulp, all-ones mantissa, 1.5 in a few binades
        for m in (0x000000, 0x000001, 0x7FFFFF, 0x400000, 0x555555, 0x2AAAAA, 0x7FFFFE):
            edge.append(np.uint32(((e + 127) << 23) | m).view(np.float32))
    z = np.concatenate([z, np.asarray(edge, np.float32), np.asarray([1e-4], np.float32)])
    assert ((z >= np.float32(1e-4)) & (z < 2.0 ** 40)).all()
    n, first = chk(0, z, (-126, -110, -103, -100, -80, -40, -14, -3, -1, 0, 1, 5, 20, 39))
    assert n == 0, first


def test_division_by_mu(chk):
    n, first = chk(1, (0.1, 0.008, 0.02, 0.05, 0.04, 1.0, 2.0 ** -30, 2.0 ** 30, 0.3333333, 7.77e-3))
    assert n == 0, first


def test_sqrt_of_the_ray_factor(chk):
    n, first = chk(2)
    assert n == 0, first


@pytest.mark.parametrize("field_name,mu", [("sdf", 0.1), ("ofusion", 0.02)])
def test_fast_and_ieee_sweeps_give_the_same_map(field_name, mu, monkeypatch):
    import torch
    from supereight_amd.pipeline import OFUSION, SDF, DenseSLAMPipeline
    from supereight_amd.synthetic import StressStream, to_colmajor
    W, H, N, dim, frames = 320, 240, 512, 4.8, 24
    field = SDF if field_name == "sdf" else OFUSION
    s = StressStream(W, H, dim)
    dev = torch.from_numpy(np.stack([s.depth(f) for f in range(frames)])).cuda()
    k = np.ascontiguousarray(s.k, np.float32)

    def run(ieee):
        if ieee:
            monkeypatch.setenv("SE_HIP_IEEE_SWEEP", "1")
        else:
            monkeypatch.delenv("SE_HIP_IEEE_SWEEP", raising=False)
        p = DenseSLAMPipeline((W, H), N, dim, field_type=field)      # (the knob is read once, here)
        for f in range(frames):
            p.frame(dev[f].data_ptr(), to_colmajor(s.pose(f)), k, mu, f)
        out = p.blocks() + p.nodes()
        p.close()
        return out

    a, b = run(False), run(True)
    assert len(a[0]) > 3000
    for u, v in zip(a, b):
        assert u.shape == v.shape and (u.view(np.uint8) == v.view(np.uint8)).all()
