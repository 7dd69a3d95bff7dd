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
