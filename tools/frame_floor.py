"""Diagnostic: wall-clock per frame of sweep + raycast only (no allocation scan, no side stream, no events)
against the full frame -- the floor a single-queue schedule could reach."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from supereight_amd.pipeline import DenseSLAMPipeline, SDF
from supereight_amd.synthetic import SyntheticStream
W, H, N, dim, mu, F = 640, 480, 512, 4.8, 0.1, 214
s = SyntheticStream(W, H, dim)
depth = torch.from_numpy(np.stack([s.depth(f) for f in range(F)])).cuda()
poses = [s.pose(f) for f in range(F)]
p = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
for f in range(14):
    p.set_depth_device(depth[f].data_ptr()); p.setPose(poses[f]); p.integration(s.k, 1, mu, f); p.raycasting(s.k, mu, f)
p.sync()
for name in ("sweep + raycast only", "full frame"):
    t0 = time.perf_counter()
    for f in range(14, F):
        p.set_depth_device(depth[f].data_ptr()); p.setPose(poses[f])
        if name == "full frame": p.integration(s.k, 1, mu, f)
        else: p.integrate_sweep(s.k, 1, mu, f)
        p.raycasting(s.k, mu, f)
    p.sync()
    t1 = time.perf_counter()
    print(f"{name:>22}: {1e6 * (t1 - t0) / (F - 14):6.1f} us/frame")
