"""Diagnostic: host time of each C-ABI call of the split frame path (alloc_scan / integrate_sweep / raycast), pipelined
(no sync), on the handle's own stream and on torch's current (null) stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from supereight_amd.pipeline import DenseSLAMPipeline, SDF
from supereight_amd.synthetic import SyntheticStream
W, H, N, dim, mu, F = 640, 480, 512, 4.8, 0.1, 210
s = SyntheticStream(W, H, dim)
depth = torch.from_numpy(np.stack([s.depth(f) for f in range(F)])).cuda()
poses = [s.pose(f) for f in range(F)]
k4 = np.ascontiguousarray(s.k, np.float32)
for mode in ("own stream", "torch null stream", "torch side stream"):
    p = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
    keep = None
    if mode == "torch null stream":
        p.set_stream(torch.cuda.current_stream().cuda_stream)
    elif mode == "torch side stream":
        keep = torch.cuda.Stream(); p.set_stream(keep.cuda_stream)
    t = {"set": 0.0, "scan": 0.0, "sweep": 0.0, "ray": 0.0}
    for f in range(F):
        if f == 10:
            p.sync(); t = {k: 0.0 for k in t}; t0 = time.perf_counter()
        a = time.perf_counter(); p.set_depth_device(depth[f].data_ptr()); p.setPose(poses[f]); b = time.perf_counter()
        p.alloc_scan(k4, 1, mu, f); c = time.perf_counter()
        p.integrate_sweep(k4, 1, mu, f); d = time.perf_counter()
        p.raycasting(k4, mu, f); e = time.perf_counter()
        t["set"] += b - a; t["scan"] += c - b; t["sweep"] += d - c; t["ray"] += e - d
    t1 = time.perf_counter(); p.sync(); t2 = time.perf_counter()
    n = F - 10
    print(mode, {k: round(1e6 * v / n, 1) for k, v in t.items()}, "host %.1f us/frame, complete %.1f us/frame" % (1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
    p.close()
