#!/usr/bin/env python3
"""CPU-only analysis (oracle built with -DSO_DIAG_MARCH, see oracle/se_oracle.cpp MarchDiag): per ray, the number of dependent
memory round trips of the SDF march under speculation depth D in unobserved space (D = 2 / 4 / 8 / 16; 2 elsewhere), aggregated
per 8x8-pixel wave tile (a wave lasts as long as its slowest ray).  usage: march_policy.py <lib> [res] [frames] [stream]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import binding  # noqa: E402
from supereight_amd.synthetic import make_stream  # noqa: E402

lib = binding._declare(C.CDLL(sys.argv[1]))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 12
kind = sys.argv[4] if len(sys.argv) > 4 else "room"
W, H, dim, mu = 640, 480, 4.8, 0.1
binding._LIBS["portable"] = lib
o = binding.OraclePipeline(binding.SDF, N, dim, W, H)
s = make_stream(kind, W, H, dim)
diag = np.zeros((H, W, 8), np.int32)
for f in range(frames):
    d, p = s.depth(f), s.pose(f)
    o.integrate(d, p, s.k, mu, f)
    if f == frames - 1:
        lib.so_set_ray_diag.argtypes = [C.c_void_p]
        lib.so_set_ray_diag(diag.ctypes.data)
    o.raycast(p, s.k, mu, f)
lib.so_set_ray_diag(None)
trips, gets, interps, y0 = diag[..., 0], diag[..., 1], diag[..., 2], diag[..., 3]
print(f"{kind} {W}x{H} -> {N}^3, frame {frames - 1}: gets/ray mean {gets.mean():.2f} max {gets.max()}, of which y==0 {y0.sum() / max(1, gets.sum()):.2%}; iterator trips mean {trips.mean():.1f} max {trips.max()}")
tiles = lambda a: a.reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
for i, D in enumerate((2, 4, 8, 16)):
    b = diag[..., 4 + i]
    tb = tiles(b)
    cost = tiles(trips + 5 * b)
    print(f"  D_empty={D:2d}: round trips/ray mean {b.mean():.2f}  p99 {np.percentile(b, 99):.0f}  max {b.max()};  per tile (slowest ray): mean {tb.mean():.2f} p90 {np.percentile(tb, 90):.0f} p99 {np.percentile(tb, 99):.0f} max {tb.max()};"
          f"  tile cost (trips + 5 x round trips) mean {cost.mean():.1f} p99 {np.percentile(cost, 99):.0f} max {cost.max()}")
