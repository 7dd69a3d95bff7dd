#!/usr/bin/env python3
"""Per-stage prediction of the 1 -> 8 GPU curve from single-GPU measurements.

No multi-GPU node is available to the builder, so the curve cannot be measured; what CAN be measured on one MI355X is every
stage a rank of an R-rank run executes, alone on the GPU as it would be on its own device:
  scan(rows of rank r), commit(the frame's complete key list), replicated sweep, raycast(rows of rank r).
For R in {2, 4, 8} and ranks {0, R/2} a row-sharded replica is driven through the frames; the peers' key lists are supplied
by a full-image helper pipeline (its scan finds every new key of the frame), so the replica ends each frame with the single
pipeline's block set.  Stage times are HIP-event averages over the timed frames.

Model of the frame period of rank r (stream plan of ShardedPipeline: scan + all-gather + commit of frame f+1 on the exchange
stream beside the raycast of frame f on the main stream; supereight_amd/multi_gpu.py):
    period_r = max( sweep + max(raycast_r, scan_r + allgather + commit) + gaps,  host )
    fps(R)   = 1 / max_r period_r
with gaps = 7 us (dispatch gaps measured on one GPU), host = 55 us (host cost of issuing a sharded frame incl. the
collective call, tools/host_overhead.py), allgather = 15 us (ASSUMED: one latency-bound RCCL all-gather of <= 128 KB per rank
over xGMI; not measurable here).  The first hardware curve can be checked against the per-stage columns.
usage: scale_predict.py [--cfg 512|2048] [--frames N] [--out path.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GAPS_US, HOST_US, ALLGATHER_US = 7.0, 55.0, 15.0


def main():
    import torch
    from supereight_amd.multi_gpu import row_partition
    from supereight_amd.pipeline import DenseSLAMPipeline, SDF
    from supereight_amd.synthetic import SyntheticStream
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="512")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    W, H, N = (640, 480, 512) if args.cfg == "512" else (1280, 960, 2048)
    dim, mu = 4.8, 0.1
    warm = 8
    n = args.frames or (60 if N == 512 else 20)
    s = SyntheticStream(W, H, dim)
    depth = torch.from_numpy(np.stack([s.depth(f) for f in range(warm + n)])).cuda()
    poses = [s.pose(f) for f in range(warm + n)]
    k = s.k
    words = 1 << 22
    send = torch.zeros(words, dtype=torch.int64, device="cuda")

    def run(rows):
        helper = DenseSLAMPipeline((W, H), N, dim, field_type=SDF, max_blocks=(1 << 16) if N == 512 else (1 << 21))   # pooled: small
        helper.set_new_keys_buffer(send.data_ptr(), words, keepalive=send)
        p = DenseSLAMPipeline((W, H), N, dim, field_type=SDF, rows=rows)
        for f in range(warm + n):
            for q in (helper, p):
                q.set_depth_device(depth[f].data_ptr()); q.setPose(poses[f])
            helper.alloc_scan(k, 1, mu, f); helper.sync()
            if f == warm:
                p.sync(); p.enable_timing(True)
            p.alloc_scan(k, 1, mu, f)
            p.alloc_commit(send.data_ptr(), 1, words)
            p.integrate_sweep(k, 1, mu, f)
            p.raycasting(k, mu, f)
            p.sync()
            helper.integrate_sweep(k, 1, mu, f); helper.sync()
        tm = p.timings(reset=True)
        nb = p.counts()[0]
        assert nb == helper.counts()[0], (nb, helper.counts())      # the replica holds the complete block set
        p.close(); helper.close()
        return {kk: round(1e3 * v["ms_sum"] / v["launches"], 2) for kk, v in tm.items() if v["launches"]}, nb

    out = {"workload": f"synthetic room+sphere {W}x{H} -> {N}^3, frames {warm}..{warm + n - 1}", "assumed_us": {"gaps": GAPS_US, "host": HOST_US, "allgather": ALLGATHER_US}, "R": {}}
    single, nb = run((0, H))
    t0 = time.time()
    for R in (1, 2, 4, 8):
        parts = row_partition(H, R)
        ranks = sorted({0, R // 2})
        rec = {"rows": {}, "stages_us": {}}
        worst = 0.0
        for r in ranks:
            st = single if R == 1 else run(parts[r])[0]
            scan, commit, sweep, ray = st.get("alloc_scan", 0.0), st.get("alloc_commit", 0.0), st.get("integrate", 0.0), st.get("raycast", 0.0)
            if R == 1:
                commit = 0.0          # a single GPU has nobody's keys to insert
            xg = 0.0 if R == 1 else ALLGATHER_US
            host = 36.0 if R == 1 else HOST_US
            period = max(sweep + max(ray, scan + xg + commit) + GAPS_US, host)
            rec["rows"][str(r)] = list(parts[r]); rec["stages_us"][str(r)] = dict(st, predicted_period_us=round(period, 1))
            worst = max(worst, period)
        rec["predicted_fps"] = round(1e6 / worst, 1)
        out["R"][str(R)] = rec
        print(f"R={R}: " + "; ".join(f"rank {r} rows {rec['rows'][r]} {rec['stages_us'][r]}" for r in rec["rows"]) + f" -> predicted {rec['predicted_fps']} fps", flush=True)
    out["blocks"] = nb
    base = out["R"]["1"]["predicted_fps"]
    for R in out["R"]:
        out["R"][R]["predicted_speedup"] = round(out["R"][R]["predicted_fps"] / base, 2)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)
    print(json.dumps({R: (v["predicted_fps"], v["predicted_speedup"]) for R, v in out["R"].items()}))


if __name__ == "__main__":
    main()
