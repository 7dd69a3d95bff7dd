#!/usr/bin/env python3
"""Same-box A/B of library builds (gpurun_ab/<name>.so via SE_HIP_LIB; "default" = the in-tree library): for each build and
workload, one subprocess runs a pipelined stream from device-resident depth frames and prints
  fps (wall clock, no events), per-kernel averages (HIP events, second pass on a fresh map), stand-alone raycast time,
  and a SHA-1 over the final map + raycast images -- every build must print the same hash (results are bit-identical).
usage: lib_ab.py [--cfgs sdf512,sdf1024,...] [--frames N] name1 name2 ...      (name = library[@ENV=VALUE...])"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = {"sdf512": (640, 480, 512, "sdf", 0.1, 60), "sdf1024": (640, 480, 1024, "sdf", 0.1, 50), "of512": (640, 480, 512, "ofusion", 0.008, 40),
       "sdf2048": (1280, 960, 2048, "sdf", 0.1, 24), "stress512": (640, 480, 512, "sdf", 0.1, 60), "stress1024": (640, 480, 1024, "sdf", 0.1, 50),
       "pooled512": (640, 480, 512, "sdf", 0.1, 60), "pooled1024": (640, 480, 1024, "sdf", 0.1, 50), "pooled2048": (1280, 960, 2048, "sdf", 0.1, 24),
       "ofstress512": (640, 480, 512, "ofusion", 0.008, 60), "pooledstress512": (640, 480, 512, "sdf", 0.1, 60), "pooledstress1024": (640, 480, 1024, "sdf", 0.1, 50), "pooledof512": (640, 480, 512, "ofusion", 0.008, 40)}


def frames_of(cfg, n):
    import numpy as np
    from supereight_amd.synthetic import make_stream
    W, H, N, field, mu, _ = CFG[cfg]
    kind = "stress" if "stress" in cfg else "room"
    path = f"/tmp/lib_ab_{kind}_{W}x{H}_{n}.npz"
    if os.path.exists(path):
        z = np.load(path)
        return z["depth"], z["poses"], z["k"]
    s = make_stream(kind, W, H, 4.8)
    depth = np.stack([s.depth(f) for f in range(n)])
    poses = np.stack([s.pose(f) for f in range(n)])
    np.savez(path, depth=depth, poses=poses, k=s.k)
    return depth, poses, s.k


def child(cfg, n):
    import numpy as np
    import torch
    from supereight_amd.pipeline import DenseSLAMPipeline, OFUSION, SDF
    from supereight_amd.synthetic import to_colmajor
    W, H, N, field, mu, _ = CFG[cfg]
    depth, poses, k = frames_of(cfg, n)
    dev = torch.from_numpy(depth).cuda()
    ptrs = [dev[f].data_ptr() for f in range(n)]
    pcm = [to_colmajor(poses[f]) for f in range(n)]
    k32 = np.ascontiguousarray(k, np.float32)
    fld = SDF if field == "sdf" else OFUSION
    kw = {}
    if cfg.startswith("pooled"):
        kw["max_blocks"] = {512: 1 << 16, 1024: 1 << 19}.get(N, 1 << 21)
    out = {"cfg": cfg, "lib": os.environ.get("SE_HIP_LIB", "default").split("/")[-1]}
    import gc
    gc.collect(); gc.freeze()   # (profiles/r04f_stall_attribution.md)
    warm = 10
    # clock ramp: keep the GPU busy ~150 ms on a throw-away map
    kw["streaming"] = True    # the one-queue schedule, as bench.py's headline (a per-frame sync flushes: the closed leg is the eager order)
    p = DenseSLAMPipeline((W, H), N, 4.8, field_type=fld, **kw)
    t0 = time.perf_counter()
    f = 0
    while time.perf_counter() - t0 < 0.15:
        for _ in range(8):
            p.frame(ptrs[f % warm], pcm[f % warm], k32, mu, 4 + f % 4)
            f += 1
        p.sync()
    p.close()
    for leg in ("wall", "closed", "closed_events", "events"):
        p = DenseSLAMPipeline((W, H), N, 4.8, field_type=fld, **kw)
        for f in range(warm):
            p.frame(ptrs[f], pcm[f], k32, mu, f)
        p.sync()
        if leg in ("events", "closed_events"):
            p.enable_timing(True)
        t0 = time.perf_counter()
        for f in range(warm, n):
            p.frame(ptrs[f], pcm[f], k32, mu, f)
            if leg in ("closed", "closed_events"):
                p.sync()
        p.sync()
        dt = time.perf_counter() - t0
        if leg == "wall":
            out["fps"] = round((n - warm) / dt, 1)
            out["us_per_frame"] = round(1e6 * dt / (n - warm), 2)
        elif leg == "closed":
            out["closed_loop_fps"] = round((n - warm) / dt, 1)
        elif leg == "closed_events":   # one sync per frame: every launch is a stand-alone kernel in the cache state the pipeline leaves (r06)
            tm = p.timings(reset=True)
            p.enable_timing(False)
            out["closed_kernels_us"] = {kk: round(1e3 * v["ms_sum"] / v["launches"], 2) for kk, v in tm.items() if v["launches"]}
        else:
            tm = p.timings(reset=True)
            p.enable_timing(False)
            out["kernels_us"] = {kk: round(1e3 * v["ms_sum"] / v["launches"], 2) for kk, v in tm.items() if v["launches"]}
            # stand-alone raycast of the last pose (nothing else on the chip)
            res = []
            for rep in range(3):
                p.enable_timing(True)
                for _ in range(10):
                    p.raycasting(k32, mu, n - 1)
                tt = p.timings(reset=True)["raycast"]
                p.enable_timing(False)
                res.append(round(1e3 * tt["ms_sum"] / tt["launches"], 2))
            out["raycast_alone_us"] = res
            h = hashlib.sha1()
            v, nn = p.vertex_normal()
            h.update(v.tobytes()); h.update(nn.tobytes())
            c, x, y, a = p.blocks()
            h.update(c.tobytes()); h.update(x.tobytes()); h.update(y.tobytes()); h.update(a.tobytes())
            out["blocks"] = int(len(c))
            out["sha1"] = h.hexdigest()[:16]
        p.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--child":
        child(a[1], int(a[2]))
        sys.exit(0)
    cfgs = ["sdf512", "sdf1024"]
    nframes = 0
    while a and a[0].startswith("--"):
        if a[0] == "--cfgs":
            cfgs = a[1].split(",")
        elif a[0] == "--frames":
            nframes = int(a[1])
        a = a[2:]
    for cfg in cfgs:
        n = nframes or CFG[cfg][5]
        frames_of(cfg, n)      # generate once, the children load the cache
        ref = None
        for name in a:
            env = dict(os.environ)
            env.pop("SE_HIP_LIB", None)
            lib_name, *knobs = name.split("@")          # "default@SE_HIP_IEEE_SWEEP=1": a library plus environment knobs
            for kv in knobs:
                kk, _, vv = kv.partition("=")
                env[kk] = vv
            if lib_name != "default":
                env["SE_HIP_LIB"] = os.path.join(ROOT, "gpurun_ab", lib_name + ".so")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", cfg, str(n)], env=env, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(f"{cfg} {name}: FAILED rc={r.returncode} {r.stderr[-600:]}", flush=True)
                continue
            d = json.loads(line[-1])
            ref = ref or d["sha1"]
            print(f"{cfg:>10} {name:>14}: {d['fps']:>8} fps ({d['us_per_frame']} us)  closed {d['closed_loop_fps']:>8} {d.get('closed_kernels_us')}  kernels {d['kernels_us']}  raycast alone {d['raycast_alone_us']}  "
                  f"blocks {d['blocks']}  sha1 {d['sha1']} {'SAME' if d['sha1'] == ref else 'DIFFERENT'}", flush=True)
