"""Diagnostic (GPU): pipelined frames (no per-frame sync), wall time per group of 20 frames, for different depth sources."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from supereight_amd.pipeline import DenseSLAMPipeline, SDF
from supereight_amd.synthetic import SyntheticStream

mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 420
W, H, N, dim, mu = 640, 480, 512, 4.8, 0.1
s = SyntheticStream(W, H, dim)
G = min(F, 100)   # frames generated; longer runs ping-pong over them (the map content does not matter here)
pp = lambda f: (f % (2 * G - 2)) if (f % (2 * G - 2)) < G else (2 * G - 2 - f % (2 * G - 2))
gen = np.stack([s.depth(f) for f in range(G)])
gposes = [s.pose(f) for f in range(G)]
class _H:
    def __getitem__(self, f):
        if isinstance(f, slice):
            return np.stack([gen[pp(i)] for i in range(f.start, min(f.stop, F))])
        return gen[pp(f)]
host = _H()
poses = [gposes[pp(f)] for f in range(F)]
dev = torch.device("cuda", 0)
depth = torch.from_numpy(gen).to(dev)
ptr = lambda f: depth[pp(f)].data_ptr()
p = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
p.set_stream(torch.cuda.current_stream(dev).cuda_stream)
for f0 in range(0, F, 20):
    t0 = time.perf_counter()
    for f in range(f0, min(F, f0 + 20)):
        if mode == "upload":
            p.set_depth(host[f])
        else:
            p.set_depth_device(ptr(f))
        p.setPose(poses[f])
        p.integration(s.k, 1, mu, f)
        p.raycasting(s.k, mu, f)
    p.sync()
    dt = time.perf_counter() - t0
    if 1e6 * dt / 20 > 100 or f0 % 400 == 0:
        print(f"{mode} frames {f0:4d}-{f0 + 19:4d}: {1e6 * dt / 20:7.1f} us/frame", flush=True)
