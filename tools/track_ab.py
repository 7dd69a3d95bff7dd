#!/usr/bin/env python3
"""Same-box A/B of the tracked loop (tracking -> integration -> raycasting, ICP poses; se_hip_frame_tracked + one sync per frame) for library builds:
usage: track_ab.py [--frames N] name1 name2 ...   (name = gpurun_ab/<name>.so via SE_HIP_LIB, "default" = the in-tree library).
Prints frames/s, the last frame's ICP iteration count and a SHA-1 over the poses and the final map: every build must print the same hash."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n):
    import numpy as np
    import torch
    from supereight_amd.pipeline import DenseSLAMPipeline
    from supereight_amd.synthetic import make_stream, to_colmajor
    W, H, N, dim, mu = 640, 480, 512, 4.8, 0.1
    s = make_stream("room", W, H, dim)
    warm = 10
    depth = np.stack([s.depth(f) for f in range(warm + n)])
    dev = torch.from_numpy(depth).cuda()
    k = np.ascontiguousarray(s.k, np.float32)
    out = {"lib": os.environ.get("SE_HIP_LIB", "default").split("/")[-1]}
    for rep in range(3):
        p = DenseSLAMPipeline((W, H), N, dim)
        h = hashlib.sha1()
        iters = 0
        for f in range(warm + n):
            if f == warm:
                p.sync(); t0 = time.perf_counter()
            if f > 3:
                p.frame_tracked(dev[f].data_ptr(), k, mu, f)
            else:
                p.setPose(s.pose(f))
                p.set_depth_device(dev[f].data_ptr())
                p.integration(k, 1, mu, f)
                p.raycasting(k, mu, f)
            h.update(np.ascontiguousarray(p.getPose()).tobytes())
        dt = time.perf_counter() - t0
        for a in p.blocks():
            h.update(np.ascontiguousarray(a).tobytes())
        out.setdefault("fps", []).append(round(n / dt, 1))
        out["iterations_last_frame"] = p.track_data()[2]
        out["sha1"] = h.hexdigest()[:16]
        p.close()
    print(json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]))
        sys.exit(0)
    n = 100
    names = sys.argv[1:]
    if names[0] == "--frames":
        n = int(names[1]); names = names[2:]
    for rnd in range(2):
        for name in names:
            env = dict(os.environ)
            if name != "default":
                env["SE_HIP_LIB"] = os.path.join(ROOT, "gpurun_ab", name + ".so")
            r = subprocess.run([sys.executable, __file__, "--child", str(n)], env=env, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            print(name, line[-1] if line else r.stderr[-600:], flush=True)
