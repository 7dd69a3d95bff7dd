#!/usr/bin/env python3
"""Follow-up to stall_probe.py: is the one long frame call tied to the number of launches since process start, to the idle -> busy
transition, or to the stream?  One process, several legs of `frames` pipelined frames each; prints where each leg's longest call sits.
  leg 1  fresh process, after `idle` s of idle
  leg 2  same process, NEW pipeline (new streams), again after `idle` s of idle
  leg 3  same process, new pipeline, NO idle in front
SE_PROBE_DUMMY=n: n empty torch launches (another stream) before leg 1."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from supereight_amd.pipeline import DenseSLAMPipeline, SDF  # noqa: E402
from supereight_amd.synthetic import StressStream, to_colmajor  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 700
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
W, H, N, PATH = 320, 240, 512, 360
s = StressStream(W, H, 4.8)
depth = torch.from_numpy(np.stack([s.depth(f) for f in range(PATH)])).cuda()
poses = [to_colmajor(s.pose(f)) for f in range(PATH)]
k = np.ascontiguousarray(s.k, np.float32)
dummy = int(os.environ.get("SE_PROBE_DUMMY", "0"))
if dummy:
    z = torch.zeros(64, device="cuda")
    t0 = time.perf_counter()
    worst = 0.0
    for i in range(dummy):
        ta = time.perf_counter(); z.add_(1.0); worst = max(worst, time.perf_counter() - ta)
    torch.cuda.synchronize()
    print(json.dumps({"dummy_launches": dummy, "ms": round(1e3 * (time.perf_counter() - t0), 2), "longest_launch_ms": round(1e3 * worst, 2)}))


def leg(name, wait):
    p = DenseSLAMPipeline((W, H), N, 4.8, field_type=SDF)
    for f in range(8):
        p.frame(depth[f].data_ptr(), poses[f], k, 0.1, f)
    p.sync()
    if wait:
        time.sleep(wait)
    st = np.zeros(frames + 1)
    st[0] = time.perf_counter()
    for f in range(frames):
        p.frame(depth[(8 + f) % PATH].data_ptr(), poses[(8 + f) % PATH], k, 0.1, 8 + f)
        st[f + 1] = time.perf_counter()
    p.sync()
    te = time.perf_counter()
    dt = np.diff(st)
    w = int(dt.argmax())
    print(json.dumps({"leg": name, "idle_before_s": wait, "fps": round(frames / (te - st[0]), 1), "median_call_us": round(1e6 * float(np.median(dt)), 1),
                      "longest_call": {"frame": w, "at_ms": round(1e3 * (st[w] - st[0]), 2), "ms": round(1e3 * float(dt[w]), 2)},
                      "calls_over_0.5ms": [(int(i), round(1e3 * float(dt[i]), 2)) for i in np.nonzero(dt > 5e-4)[0][:10]]}), flush=True)
    p.close()


leg("1 fresh process", idle)
leg("2 new pipeline after idle", idle)
leg("3 new pipeline, no idle", 0.0)
