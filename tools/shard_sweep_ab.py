"""A/B of the sharded sweep (SURVEY 8(e) option 4) on ONE GPU: R row-sharded replicas of one process, the all-gathers
emulated by device concatenation.  Per R: time of the integration kernel per replica, replicated vs owner-computes, the
apply kernel, and the bytes a replica would receive over xGMI per frame.   RES=512|1024 FIELD=sdf R_LIST=2,4,8"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from supereight_amd.multi_gpu import row_partition
from supereight_amd.pipeline import DenseSLAMPipeline, SDF, OFUSION
from supereight_amd.synthetic import SyntheticStream

W, H, N, dim = 640, 480, int(os.environ.get("RES", 512)), 4.8
FIELD = OFUSION if os.environ.get("FIELD", "sdf") == "ofusion" else SDF
mu = float(os.environ.get("MU", 0.1 if FIELD == SDF else 0.008))
FRAMES, TIMED_FROM = 16, 6
dev = torch.device("cuda", 0)
stream = SyntheticStream(W, H, dim)
depths = [stream.depth(f) for f in range(FRAMES)]
poses = [stream.pose(f) for f in range(FRAMES)]
out = {"config": {"W": W, "H": H, "res": N, "field": "ofusion" if FIELD == OFUSION else "sdf", "mu": mu, "frames_timed": FRAMES - TIMED_FROM}, "runs": []}


def run(R, sharded):
    parts = row_partition(H, R)
    reps = [DenseSLAMPipeline((W, H), N, dim, field_type=FIELD, rows=parts[r]) for r in range(R)]
    words = 1 << 20
    cap = 64 * ((int(2.5 * 20000 * (N // 512) ** 2 / R) + 63) // 64)
    send = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(R)]
    seg = reps[0].sweep_shard_bytes(cap)
    bsend = [torch.zeros(seg, dtype=torch.uint8, device=dev) for _ in range(R)] if sharded else None
    for r, p in enumerate(reps):
        p.set_new_keys_buffer(send[r].data_ptr(), words, keepalive=send[r])
        if sharded:
            p.set_sweep_shard(r, R, bsend[r].data_ptr(), cap, keepalive=bsend[r])
    recs = bricks = 0
    for f in range(FRAMES):
        if f == TIMED_FROM:
            for p in reps:
                p.sync(); p.enable_timing(True); p.timings(reset=True)
        for p in reps:
            p.set_depth(depths[f]); p.setPose(poses[f])
            p.alloc_scan(stream.k, 1, mu, f)
        for p in reps:
            p.sync()
        recv = torch.cat(send); torch.cuda.synchronize()
        for p in reps:                      # one replica at a time: its kernels have the GPU to themselves, as on its own GPU
            p.alloc_commit(recv.data_ptr(), R, words)
            p.integrate_sweep(stream.k, 1, mu, f)
            p.sync()
        if sharded:
            brecv = torch.cat(bsend); torch.cuda.synchronize()
            if f >= TIMED_FROM:
                for b in bsend:
                    cnt = b[:512].view(torch.int64).cpu().numpy()
                    r_ = b[512:512 + 4 * cap].view(torch.int32).cpu().numpy().reshape(64, cap // 64)
                    for sub in range(64):
                        recs += int(cnt[sub]); bricks += int((r_[sub, :cnt[sub]] < 0).sum())
            for p in reps:
                p.apply_bricks(brecv.data_ptr(), R)
                p.sync()
        for p in reps:
            p.raycasting(stream.k, mu, f)
            p.sync()
    t = [p.timings(reset=True) for p in reps]
    nt = FRAMES - TIMED_FROM
    avg = lambda k: [1e3 * x[k]["ms_sum"] / max(1, x[k]["launches"]) for x in t]
    res = {"R": R, "sharded_sweep": sharded, "integrate_us_per_replica": [round(v, 1) for v in avg("integrate")],
           "apply_us_per_replica": [round(v, 1) for v in avg("apply_bricks")] if sharded else None,
           "raycast_us_per_replica": [round(v, 1) for v in avg("raycast")]}
    if sharded:
        per_frame_bricks = bricks / nt
        res["bricks_per_frame_all_ranks"] = round(per_frame_bricks)
        res["records_per_frame_all_ranks"] = round(recs / nt)
        res["payload_MB_received_per_replica"] = round(per_frame_bricks * 4096 * (R - 1) / R / 1e6, 2)
        res["allgather_MB_received_per_replica_fixed_capacity"] = round(seg * (R - 1) / 1e6, 1)
        res["xgmi_us_at_1TBps_payload"] = round(per_frame_bricks * 4096 * (R - 1) / R / 1e12 * 1e6, 1)
    nb = reps[0].counts()[0]
    res["blocks"] = nb
    for p in reps:
        p.close()
    return res


for R in [int(x) for x in os.environ.get("R_LIST", "2,4,8").split(",")]:
    for sharded in (False, True):
        r = run(R, sharded)
        out["runs"].append(r)
        print(json.dumps(r), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/shard_sweep_ab_{N}.json", "w"), indent=1)
