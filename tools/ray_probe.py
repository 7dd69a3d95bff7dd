#!/usr/bin/env python3
"""Closed-loop runner for profiling passes: one se_hip_sync per frame, so every launch of the trace is a stand-alone kernel in the
cache state the pipeline leaves it in (the raycast right behind its frame's sweep) -- not the warm repeat of `raycast alone` in tools/lib_ab.py.
  ray_probe.py <cfg of lib_ab.py> [frames] [--streaming]     (--streaming: no sync between frames -> the fused k_raycast_scan launches instead)
Prints one JSON line (frames/s of the loop, blocks)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import numpy as np
    import torch
    from lib_ab import CFG, frames_of
    from supereight_amd.pipeline import DenseSLAMPipeline, OFUSION, SDF
    from supereight_amd.synthetic import to_colmajor
    cfg = sys.argv[1]
    args = [a for a in sys.argv[2:] if not a.startswith("--")]
    streaming = "--streaming" in sys.argv
    W, H, N, field, mu, n0 = CFG[cfg]
    n = int(args[0]) if args else n0
    depth, poses, k = frames_of(cfg, n)
    dev = torch.from_numpy(depth).cuda()
    ptrs = [dev[f].data_ptr() for f in range(n)]
    pcm = [to_colmajor(poses[f]) for f in range(n)]
    k32 = np.ascontiguousarray(k, np.float32)
    kw = {}
    if cfg.startswith("pooled"):
        kw["max_blocks"] = {512: 1 << 16, 1024: 1 << 19}.get(N, 1 << 21)
    p = DenseSLAMPipeline((W, H), N, 4.8, field_type=SDF if field == "sdf" else OFUSION, streaming=streaming, **kw)
    warm = 10
    for f in range(warm):
        p.frame(ptrs[f], pcm[f], k32, mu, f)
    p.sync()
    t0 = time.perf_counter()
    for f in range(warm, n):
        p.frame(ptrs[f], pcm[f], k32, mu, f)
        if not streaming:
            p.sync()
    p.sync()
    dt = time.perf_counter() - t0
    print(json.dumps({"cfg": cfg, "streaming": streaming, "fps": round((n - warm) / dt, 1), "blocks": int(p.counts()[0]), "lib": os.environ.get("SE_HIP_LIB", "default").split("/")[-1]}))
    p.close()


if __name__ == "__main__":
    main()
