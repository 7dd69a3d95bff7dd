#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).

The reference cannot be built in this image (Eigen3 / Sophus absent), so these are ORACLE outputs,
not reference outputs: they pin the oracle against accidental change and give the GPU tests a
fixture that needs neither the oracle nor /root/reference at run time.  Per field type: 5 frames of
the synthetic stream, 80x60 depth into a 128^3 / 2.4 m volume.  Stored: the depth frames and poses
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
from supereight_amd.synthetic import SyntheticStream  # noqa: E402

W, H, N, DIM, FRAMES = 80, 60, 128, 2.4, 5


def crc_rows(a):
    return np.array([zlib.crc32(np.ascontiguousarray(r).tobytes()) for r in a], np.uint32)


def main():
    for name, field, mu in (("sdf", SDF, 0.1), ("ofusion", OFUSION, 0.02)):
        s = SyntheticStream(W, H, DIM)
        o = OraclePipeline(field, N, DIM, W, H)
        depths, poses = [], []
        for f in range(FRAMES):
            d, p = s.depth(f), s.pose(f)
            depths.append(d)
            poses.append(p)
            o.integrate(d, p, s.k, mu, f)
            ran, v, n = o.raycast(p, s.k, mu, f)
        assert ran
        assert o.stats()["truncated"] == 0   # a saturated key buffer makes the reference itself nondeterministic
        c, x, y, a = o.blocks()
        code, side, nx, ny = o.nodes()
        out = os.path.join(ROOT, "tests", "golden", f"{name}_{W}x{H}_{N}.npz")
        np.savez_compressed(out, depth=np.stack(depths), pose=np.stack(poses), k=s.k, mu=np.float32(mu),
                            dims=np.array([W, H, N, FRAMES], np.int32), dim=np.float32(DIM),
                            coords=c, active=a, crc_x=crc_rows(x), crc_y=crc_rows(y),
                            sum_x=x.astype(np.float64).sum(1), sum_y=y.astype(np.float64).sum(1),
                            node_code=code, node_crc_x=crc_rows(nx), node_crc_y=crc_rows(ny),
                            vertex=v, normal=n)
        print(name, "blocks", len(c), "nodes", len(code), "hits", int((n[..., 0] != -2).sum()), os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
