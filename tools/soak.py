#!/usr/bin/env python3
"""Soak run (GPU): N frames of the ICL-like stress stream, its 360-frame camera path repeated, through one pipeline -- frames/s per
window, block / node counts, and that no pool, key list or counter misbehaves over a long session.  usage: soak.py [frames] [res] [W H]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from supereight_amd.pipeline import DenseSLAMPipeline, OFUSION, SDF  # noqa: E402
from supereight_amd.synthetic import StressStream, to_colmajor  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1080
import gc  # noqa: E402
gc.collect(); gc.freeze()   # the interpreter's full collection over the import-time heap otherwise lands in window 3 (r03: 2.6 k frames/s there; profiles/r04f_stall_attribution.md)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (320, 240)
PATH = 360
s = StressStream(W, H, 4.8)
depth = torch.from_numpy(np.stack([s.depth(f) for f in range(PATH)])).cuda()
poses = [to_colmajor(s.pose(f)) for f in range(PATH)]
k = np.ascontiguousarray(s.k, np.float32)
OF = os.environ.get("SE_SOAK_FIELD", "sdf") == "ofusion"       # SE_SOAK_FIELD=ofusion: occupancy mapping, mu = 0.02
MU = 0.02 if OF else 0.1
p = DenseSLAMPipeline((W, H), N, 4.8, field_type=OFUSION if OF else SDF, streaming=os.environ.get("SE_SOAK_EAGER") is None)   # the one-queue schedule unless SE_SOAK_EAGER is set
out = {"workload": f"stress stream {W}x{H} -> {N}^3 {'OFusion' if OF else 'SDF'}, {frames} frames (path of {PATH} repeated)", "windows": []}
t0 = time.perf_counter()
for f in range(frames):
    p.frame(depth[f % PATH].data_ptr(), poses[f % PATH], k, MU, f)
    if f % 120 == 119:
        p.sync()
        t1 = time.perf_counter()
        nb, nn = p.counts()      # raises on overflow
        out["windows"].append({"frames": [f - 119, f], "fps": round(120 / (t1 - t0), 1), "blocks": nb, "nodes": nn})
        t0 = time.perf_counter()
v, n = p.vertex_normal()
out["final_hits"] = int((n[..., 0] != -2).sum())
c, a = p.block_flags()
out["final_active_blocks"] = int(a.sum())
assert set(np.unique(a).tolist()) <= {0, 1}
p.close()
print(json.dumps(out))
