"""Diagnostic (GPU): period of every frame in the pipelined loop (event after each frame's raycast on the main stream,
differences of consecutive events), after the same pre-warm bench.py does."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from supereight_amd.multi_gpu import ShardedPipeline
from supereight_amd.pipeline import SDF
from supereight_amd.synthetic import SyntheticStream, to_colmajor

F = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W, H, N, dim, mu = 640, 480, 512, 4.8, 0.1
s = SyntheticStream(W, H, dim)
depth = torch.from_numpy(np.stack([s.depth(f) for f in range(F)])).cuda()
poses = [s.pose(f) for f in range(F)]
pcm = [to_colmajor(q) for q in poses]
ptrs = [depth[f].data_ptr() for f in range(F)]
k = np.ascontiguousarray(s.k, np.float32)
import types
args = types.SimpleNamespace(width=W, height=H, res=N, dim=dim, mu=mu)
sp = ShardedPipeline((W, H), N, dim, SDF, 0, 1, 0)
_, scratch = bench.prewarm(args, SDF, ptrs, poses, k, 0, ms=float(os.environ.get("PREWARM_MS", 90)))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(F)]
host = []
for f in range(F):
    t0 = time.perf_counter()
    sp.frame(ptrs[f], pcm[f], k, mu, f)
    host.append(1e6 * (time.perf_counter() - t0))
    ev[f].record()
torch.cuda.synchronize()
per = [ev[f - 1].elapsed_time(ev[f]) * 1e3 for f in range(1, F)]
print("frame periods us (frame 1..):", " ".join(f"{p:.0f}" for p in per))
print("host enqueue us:", " ".join(f"{h:.0f}" for h in host))
print("mean period frames 5..24: %.1f   25..%d: %.1f" % (np.mean(per[4:24]), F - 1, np.mean(per[24:])))
sp.close()
