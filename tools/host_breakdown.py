"""Diagnostic: host time of each call ShardedPipeline.frame makes (one-rank RCCL group, exchange on)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from supereight_amd.multi_gpu import ShardedPipeline
from supereight_amd.pipeline import SDF
from supereight_amd.synthetic import SyntheticStream
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
W, H, N, dim, mu, F = 640, 480, 512, 4.8, 0.1, 410
s = SyntheticStream(W, H, dim)
depth = torch.from_numpy(np.stack([s.depth(f) for f in range(F)])).cuda()
poses = [s.pose(f) for f in range(F)]
ptrs = [depth[f].data_ptr() for f in range(F)]
sp = ShardedPipeline((W, H), N, dim, SDF, 0, 1, 0, exchange_always=True)
for f in range(210): sp.frame(ptrs[f], poses[f], s.k, mu, f)
torch.cuda.synchronize()
p = sp.p
acc = {}
def tick(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
words = sp.small_words
send, recv = sp._views[words]
n = 0
for f in range(210, F):
    t = time.perf_counter()
    p.set_depth_device(ptrs[f]); t = tick("set_depth_device", t)
    p.setPose(poses[f]); t = tick("setPose", t)
    p.alloc_scan(s.k, 1, mu, f); t = tick("alloc_scan", t)
    w = sp._pg._allgather_base(recv, send); t = tick("allgather_base", t)
    w.wait(); t = tick("work.wait", t)
    sp.main.wait_stream(sp.xs); t = tick("wait_stream", t)
    p.alloc_commit(recv.data_ptr(), 1, words); t = tick("alloc_commit", t)
    p.integrate_sweep(s.k, 1, mu, f); t = tick("integrate_sweep", t)
    p.raycasting(s.k, mu, f); t = tick("raycasting", t)
    n += 1
    if n % 50 == 0: torch.cuda.synchronize()
torch.cuda.synchronize()
for k, v in acc.items(): print(f"{k:>18}: {1e6 * v / n:6.1f} us")
print(f"{'total':>18}: {1e6 * sum(acc.values()) / n:6.1f} us")
dist.destroy_process_group()
