#!/usr/bin/env python3
"""Diagnostic (variant build -DSE_SCHED_TIMING): where workgroup 0's raycast scheduler sits inside a sweep launch.  100 MHz realtime ticks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from supereight_amd.pipeline import DenseSLAMPipeline
from supereight_amd.synthetic import make_stream, to_colmajor
kind = sys.argv[1] if len(sys.argv) > 1 else "stress"
s = make_stream(kind, 640, 480, 4.8)
n = 40
depth = torch.from_numpy(np.stack([s.depth(f) for f in range(n)])).cuda()
k = np.ascontiguousarray(s.k, np.float32)
p = DenseSLAMPipeline((640, 480), 512, 4.8, streaming=True)
for f in range(n):
    p.frame(depth[f].data_ptr(), to_colmajor(s.pose(f)), k, 0.1, f)
    if f >= n - 6:
        p.sync()
        st = p.stats()
        t0, s_in, s_out, b_end = (~st["clk_stage"]) & ((1 << 64) - 1), st["r13"], st["r14"], st["r15"]
        print(f"frame {f}: scheduler enters {10 * (s_in - t0)} ns after the first wave, runs {10 * (s_out - s_in)} ns; last block wave ends at {10 * (b_end - t0)} ns")
        p.lib.se_hip_enable_stats(p._h, 0)     # (zeroes the stats words)
