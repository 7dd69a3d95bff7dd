"""Oracle restatement of bilateralFilterKernel (se_denseslam/src/preprocessing.cpp:41-89) against an
independent float64 numpy evaluation of the same formula (the reference's tests do not cover it)."""
import numpy as np

from oracle.binding import oracle_bilateral_filter


def _numpy_filter(d):
    H, W = d.shape
    g = np.exp(-((np.arange(5) - 2) ** 2) / 32.0)
    out = np.zeros_like(d, dtype=np.float64)
    for y in range(H):
        for x in range(W):
            c = float(d[y, x])
            if c == 0:
                continue
            t = s = 0.0
            for i in range(-2, 3):
                for j in range(-2, 3):
                    p = float(d[min(max(y + j, 0), H - 1), min(max(x + i, 0), W - 1)])
                    if p > 0:
                        fct = g[i + 2] * g[j + 2] * np.exp(-((p - c) ** 2) / (0.1 * 0.1 * 2))
                        t += fct * p; s += fct
            out[y, x] = t / s
    return out


def test_bilateral_filter_matches_formula():
    rng = np.random.default_rng(7)
    d = (rng.integers(400, 4000, size=(24, 32)).astype(np.float32) / 1000.0).astype(np.float32)
    d[rng.random(d.shape) < 0.1] = 0.0
    out = oracle_bilateral_filter(d)
    ref = _numpy_filter(d)
    assert ((out == 0) == (d == 0)).all()
    assert np.abs(out - ref).max() < 2e-6


def test_bilateral_filter_constant_image_and_edges():
    d = np.full((16, 16), 1.25, np.float32)
    assert (oracle_bilateral_filter(d) == d).all()          # weights cancel exactly: t / sum == centre
    # a depth step larger than e_delta is preserved (range kernel exp(-(0.5^2)/0.02) ~ 4e-6)
    d[:, 8:] = 1.75
    out = oracle_bilateral_filter(d)
    assert np.abs(out - d).max() < 1e-4


def test_sincos_redefinition_stays_within_a_bound_of_libm_arithmetic():
    """the parity oracle (and the device) DEFINE the sin / cos inside Sophus::SE3f::exp as the correctly rounded values,
    where the reference calls its C library's sinf / cosf (1 ulp off on a fraction of a percent of arguments).  The same closed SLAM
    loop -- tracking -> integration -> raycasting, 40 tracked frames of the room stream at 160x120 -> 128^3 -- is run with both
    definitions; the tracked poses must stay within 2e-6 m / 2e-7 per rotation entry of each other at every frame, which is what
    ties the bit-exact ICP gate of tests/test_gpu_tracking.py to the reference's arithmetic by a measured number."""
    from oracle.binding import SDF, OraclePipeline, load, oracle_tracking
    from supereight_amd.synthetic import SyntheticStream
    W, H, N, dim, mu, frames = 160, 120, 128, 2.4, 0.1, 44
    lib = load()

    def loop(libm):
        lib.so_set_libm_sincos(1 if libm else 0)
        try:
            s = SyntheticStream(W, H, dim)
            o = OraclePipeline(SDF, N, dim, W, H)
            pose = s.pose(0).copy()
            v = n = rp = None
            poses, accepted = [], 0
            for f in range(frames):
                d = s.depth(f)
                if f >= 4:
                    ok, pose, _, _, _ = oracle_tracking(d, s.k, pose, rp, v, n, 1e-5, (10, 5, 4))
                    accepted += int(ok)
                else:
                    pose = s.pose(f).copy()
                poses.append(pose.copy())
                o.integrate(d, pose, s.k, mu, f)
                ran, vv, nn = o.raycast(pose, s.k, mu, f)
                if ran:
                    v, n, rp = vv, nn, pose.copy()
            o.close()
            return np.stack(poses), accepted
        finally:
            lib.so_set_libm_sincos(0)

    a, acc_a = loop(False)
    b, acc_b = loop(True)
    assert acc_a == acc_b == frames - 4
    dt = np.abs(a[:, :3, 3] - b[:, :3, 3]).max()
    dr = np.abs(a[:, :3, :3] - b[:, :3, :3]).max()
    print(f"correctly rounded vs libm sinf/cosf over {frames - 4} tracked frames: translation {dt:.2e} m, rotation entry {dr:.2e}")
    assert dt < 2e-6 and dr < 2e-7
