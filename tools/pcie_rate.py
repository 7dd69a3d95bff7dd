"""PCIe-inclusive frame rate: the depth image is handed over as a HOST buffer every frame (the reference's
boundary: preprocessing() gets a host uint16 image), instead of being resident in HBM as in bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from supereight_amd.pipeline import DenseSLAMPipeline, SDF
from supereight_amd.synthetic import SyntheticStream, render_depth_mm
W, H, N, dim, mu, F, warm = 640, 480, 512, 4.8, 0.1, 110, 10
s = SyntheticStream(W, H, dim)
depth = [s.depth(f) for f in range(F)]
mm = [np.where(d == 0, 0, np.round(d * 1000)).astype(np.uint16) for d in depth]   # same holes, millimetres
poses = [s.pose(f) for f in range(F)]
for name in ("float32 metres (se_hip_upload_depth, 1.2 MB/frame)", "uint16 mm (se_hip_upload_depth_mm, 0.6 MB/frame, mm2meters fused)"):
    p = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
    for f in range(F):
        if f == warm:
            p.sync(); t0 = time.perf_counter()
        if name.startswith("float32"): p.set_depth(depth[f])
        else: p.set_depth_mm(mm[f])
        p.setPose(poses[f])
        p.integration(s.k, 1, mu, f)
        p.raycasting(s.k, mu, f)
    p.sync(); t1 = time.perf_counter()
    print(f"{name}: {(F - warm) / (t1 - t0):8.0f} frames/s  ({1e3 * (t1 - t0) / (F - warm):.3f} ms/frame)")
    p.close()
