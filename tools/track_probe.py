"""The reference's loop with tracking on (se_apps/src/benchmark.cpp:115-150) on one GPU: frames/s of the variants of the host side.
  python tools/track_probe.py                 A/B table: four calls per frame vs se_hip_frame_tracked, SE_HIP_ICP_LOOKAHEAD 0 / 1 / 2, with and
                                              without a device sync per frame
  python tools/track_probe.py --trace N       N tracked frames through se_hip_frame_tracked (run it under rocprofv3 --kernel-trace)
Same stream and sizes as bench.py's tracking_on leg (640x480 -> 512^3 SDF, mu 0.1, GT poses for frames 0..3)."""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(args, one_call, look, sync, n, warm=10, stream="room"):
    import torch
    from supereight_amd.pipeline import DenseSLAMPipeline
    from supereight_amd.synthetic import StressStream, SyntheticStream
    W, H, N, dim, mu = args.width, args.height, args.res, args.dim, args.mu
    s = SyntheticStream(W, H, dim) if stream == "room" else StressStream(W, H, dim, time_scale=0.25, start=36)
    dev = torch.from_numpy(np.stack([s.depth(f) for f in range(warm + n)])).cuda()
    ptrs = [dev[f].data_ptr() for f in range(warm + n)]
    k = np.ascontiguousarray(s.k, np.float32).reshape(4)
    os.environ["SE_HIP_ICP_LOOKAHEAD"] = str(look)
    p = DenseSLAMPipeline((W, H), N, dim)
    gc.collect(); gc.freeze()
    tracked = iters = 0
    t0 = None
    for f in range(warm + n):
        if f == warm:
            p.sync(); t0 = time.perf_counter()
        if f <= 3:
            p.setPose(s.pose(f)); p.set_depth_device(ptrs[f]); p.integration(k, 1, mu, f); p.raycasting(k, mu, f)
        elif one_call:
            tracked += (p.frame_tracked(ptrs[f], k, mu, f) >> 2) & 1
        else:
            p.set_depth_device(ptrs[f])
            ok = p.tracking(k, 1e-5, 1, f)
            tracked += int(ok)
            if ok:
                p.integration(k, 1, mu, f)
            p.raycasting(k, mu, f)
        if sync:
            p.sync()
    p.sync()
    dt = time.perf_counter() - t0
    err = float(np.abs(p.getPose()[:3, 3] - np.asarray(s.pose(warm + n - 1))[:3, 3]).max())
    p.close()
    return {"one_call": one_call, "lookahead": look, "sync_per_frame": sync, "stream": stream, "fps": n / dt, "us_per_frame": 1e6 * dt / n, "tracked": tracked, "of": warm + n - 4,
            "final_position_error_m": err}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=640); ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--res", type=int, default=512); ap.add_argument("--dim", type=float, default=4.8); ap.add_argument("--mu", type=float, default=0.1)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--trace", type=int, default=0)
    ap.add_argument("--full", action="store_true", help="every variant, twice (default: three variants, once)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.trace:
        print(json.dumps(run(a, True, int(os.environ.get("SE_HIP_ICP_LOOKAHEAD", "2")), True, a.trace)))
        return
    rows = []
    variants = ((False, 0, True), (False, 2, True), (True, 0, True), (True, 1, True), (True, 2, True), (True, 3, True), (True, 2, False)) if a.full else \
               ((False, 0, True), (True, 2, True), (True, 2, False))
    for rep in range(2 if a.full else 1):
        for stream in ("room", "stress"):
            for one_call, look, sync in variants:
                r = run(a, one_call, look, sync, a.frames, stream=stream)
                r["rep"] = rep
                rows.append(r)
                print(json.dumps(r), flush=True)
    if a.out:
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
