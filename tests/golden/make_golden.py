#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).

The reference cannot be built in this image (Eigen3 / Sophus absent), so these are ORACLE outputs,
not reference outputs: they pin the oracle against accidental change and give the GPU tests a
fixture that needs neither the oracle nor /root/reference at run time.  Per field type: 5 frames of
the synthetic stream, 80x60 depth into a 128^3 / 2.4 m volume -- and (r03) 8 frames of the ICL-like stress stream
(every 12th frame of its camera path: 24 deg and ~15 cm between consecutive inputs, scene clipped by the volume, sensor noise,
negative fy), whose raycasts leave the volume (stats.oob > 0, stored) without touching the reads the reference leaves undefined
(stats.oob_ub == 0).  Stored: the depth frames and poses
(inputs), the sorted block coordinates, per-block CRC32 of the x and y planes (bit-exact check
without storing 512 voxels per block), per-block float64 sums, node keys + CRC, and the last
frame's vertex / normal maps.
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import OFUSION, SDF, OraclePipeline  # noqa: E402
from supereight_amd.synthetic import StressStream, SyntheticStream  # noqa: E402

W, H, N, DIM, FRAMES = 80, 60, 128, 2.4, 5


def crc_rows(a):
    return np.array([zlib.crc32(np.ascontiguousarray(r).tobytes()) for r in a], np.uint32)


class _Strided:
    """every `stride`-th frame of a stream (which must be asked for its frames in order)"""

    def __init__(self, s, stride):
        self.s, self.stride, self.k = s, stride, s.k

    def depth(self, f):
        d = None
        for g in range(f * self.stride - (self.stride - 1) if f else 0, f * self.stride + 1):
            d = self.s.depth(g)
        return d

    def pose(self, f):
        return self.s.pose(f * self.stride)


def main():
    for name, field, mu, frames in (("sdf", SDF, 0.1, FRAMES), ("ofusion", OFUSION, 0.02, FRAMES), ("stress_sdf", SDF, 0.1, 8), ("stress_ofusion", OFUSION, 0.02, 8)):
        s = _Strided(StressStream(W, H, DIM), 12) if name.startswith("stress") else SyntheticStream(W, H, DIM)
        o = OraclePipeline(field, N, DIM, W, H)
        o.count_stats(True)
        depths, poses = [], []
        for f in range(frames):
            d, p = s.depth(f), s.pose(f)
            depths.append(d)
            poses.append(p)
            o.integrate(d, p, s.k, mu, f)
            ran, v, n = o.raycast(p, s.k, mu, f)
        assert ran
        st = o.stats()
        assert st["truncated"] == 0   # a saturated key buffer makes the reference itself nondeterministic
        assert st["oob_ub"] == 0      # no read that the reference leaves undefined
        assert st["oob"] > 0 or not name.startswith("stress"), st
        c, x, y, a = o.blocks()
        code, side, nx, ny = o.nodes()
        out = os.path.join(ROOT, "tests", "golden", f"{name}_{W}x{H}_{N}.npz")
        np.savez_compressed(out, depth=np.stack(depths), pose=np.stack(poses), k=s.k, mu=np.float32(mu),
                            dims=np.array([W, H, N, frames], np.int32), dim=np.float32(DIM), oob=np.int64(st["oob"]),
                            coords=c, active=a, crc_x=crc_rows(x), crc_y=crc_rows(y),
                            sum_x=x.astype(np.float64).sum(1), sum_y=y.astype(np.float64).sum(1),
                            node_code=code, node_crc_x=crc_rows(nx), node_crc_y=crc_rows(ny),
                            vertex=v, normal=n)
        print(name, "blocks", len(c), "nodes", len(code), "hits", int((n[..., 0] != -2).sum()), os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
