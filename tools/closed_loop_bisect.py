#!/usr/bin/env python3
"""Same-box, same-session comparison of whole source trees (where did the closed-loop figure go between the r02
and r03 driver runs?).  Each argument is a directory holding a `supereight_amd/` package with its own built libse_hip.so
(gpurun_ab/wt_<commit>/, made from `git worktree add` + build; "." = the current tree).  For every tree, in alternating order and
`--reps` times, a subprocess runs the 640x480 -> 512^3 SDF room stream from device-resident frames and prints
  closed-loop frames/s (one sync per frame, benchmark.cpp:148-167's bracketing) and pipelined frames/s (one sync at the end).
usage: closed_loop_bisect.py [--reps 3] [--frames 100] dir1 dir2 ..."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(n):
    import numpy as np
    import torch
    from supereight_amd.pipeline import DenseSLAMPipeline, SDF
    from supereight_amd.synthetic import SyntheticStream, to_colmajor
    W, H, N, mu, warm = 640, 480, 512, 0.1, 10
    path = f"/tmp/clb_{n}.npz"
    if os.path.exists(path):
        z = np.load(path); depth, poses, k = z["depth"], z["poses"], z["k"]
    else:
        s = SyntheticStream(W, H, 4.8)
        depth = np.stack([s.depth(f) for f in range(n)]); poses = np.stack([s.pose(f) for f in range(n)]); k = np.asarray(s.k)
        np.savez(path, depth=depth, poses=poses, k=k)
    dev = torch.from_numpy(depth).cuda()
    ptrs = [dev[f].data_ptr() for f in range(n)]
    pcm = [to_colmajor(poses[f]) for f in range(n)]
    k32 = np.ascontiguousarray(k, np.float32)

    def one(p, f):
        if hasattr(p, "frame"):
            p.frame(ptrs[f], pcm[f], k32, mu, f)
        else:
            p.set_depth_device(ptrs[f]); p.setPose(poses[f]); p.integration(k32, 1, mu, f); p.raycasting(k32, mu, f)

    # clock ramp: ~150 ms of the same kernels on a throw-away map
    p = DenseSLAMPipeline((W, H), N, 4.8, field_type=SDF)
    t0 = time.perf_counter(); f = 0
    while time.perf_counter() - t0 < 0.15:
        for _ in range(8):
            one(p, 4 + f % 6); f += 1
        p.sync()
    out = {}
    for leg in ("closed", "pipelined"):
        q = DenseSLAMPipeline((W, H), N, 4.8, field_type=SDF)
        for f in range(warm):
            one(q, f)
            if leg == "closed":
                q.sync()
        q.sync()
        t0 = time.perf_counter()
        for f in range(warm, n):
            one(q, f)
            if leg == "closed":
                q.sync()
        q.sync()
        out[leg] = round((n - warm) / (time.perf_counter() - t0), 1)
        q.close()
    p.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--child":
        child(int(a[1])); sys.exit(0)
    reps, frames = 3, 100
    while a and a[0].startswith("--"):
        if a[0] == "--reps": reps = int(a[1])
        elif a[0] == "--frames": frames = int(a[1])
        a = a[2:]
    res = {d: [] for d in a}
    for r in range(reps):
        for d in (a if r % 2 == 0 else a[::-1]):
            tree = ROOT if d == "." else os.path.join(ROOT, d)
            env = dict(os.environ); env["PYTHONPATH"] = tree; env.pop("SE_HIP_LIB", None)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(frames)], env=env, cwd=tree, capture_output=True, text=True)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(f"{d}: FAILED {out.stderr[-400:]}", flush=True); continue
            res[d].append(json.loads(line[-1]))
            print(f"rep {r} {d:>24}: closed {res[d][-1]['closed']:>9} pipelined {res[d][-1]['pipelined']:>9}", flush=True)
    print("\nmedian over repetitions:")
    for d in a:
        if res[d]:
            c = sorted(x["closed"] for x in res[d]); q = sorted(x["pipelined"] for x in res[d])
            print(f"{d:>24}: closed {c[len(c) // 2]:>9} pipelined {q[len(q) // 2]:>9}   (all closed: {c})")
