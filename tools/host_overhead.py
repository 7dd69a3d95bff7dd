"""Diagnostic: host time per frame of ShardedPipeline.frame (enqueue only, no device sync), with and
without the exchange step (one-rank RCCL group), against the device time per frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from supereight_amd.multi_gpu import ShardedPipeline
from supereight_amd.pipeline import SDF
from supereight_amd.synthetic import SyntheticStream, to_colmajor
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
W, H, N, dim, mu, F = 640, 480, 512, 4.8, 0.1, 210
s = SyntheticStream(W, H, dim)
depth = torch.from_numpy(np.stack([s.depth(f) for f in range(F)])).cuda()
poses = [s.pose(f) for f in range(F)]
ptrs = [depth[f].data_ptr() for f in range(F)]
poses_cm = [to_colmajor(q) for q in poses]
k4 = np.ascontiguousarray(s.k, np.float32)
print("SE_HIP_HOST_GATE =", os.environ.get("SE_HIP_HOST_GATE", "(default: on)"))
for ex, one_call in ((False, True), (False, False), (True, False), (False, True), (True, False)):
    sp = ShardedPipeline((W, H), N, dim, SDF, 0, 1, 0, exchange_always=ex)
    P, K = (poses_cm, k4) if one_call else (poses, s.k)      # (16,) poses: the whole frame is one FFI call (se_hip_frame)
    for f in range(10): sp.frame(ptrs[f], P[f], K, mu, f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(10, F): sp.frame(ptrs[f], P[f], K, mu, f)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = F - 10
    print(f"exchange={ex} one_call={one_call}: host enqueue {1e6 * (t1 - t0) / n:.1f} us/frame, device-complete {1e6 * (t2 - t0) / n:.1f} us/frame")
    sp.close()
dist.destroy_process_group()
