"""Diagnostic (GPU): raycast kernel time under different tuning-knob settings, results compared bit for bit.
usage: ray_ab.py [sdf512|sdf1024|of512|sdf2048] -- KEY=VAL[,KEY=VAL] ...   (each group = one run; "-" = defaults)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from supereight_amd.pipeline import DenseSLAMPipeline, SDF, OFUSION
from supereight_amd.synthetic import SyntheticStream

CFG = {"sdf512": (640, 480, 512, SDF, 0.1), "sdf1024": (640, 480, 1024, SDF, 0.1), "of512": (640, 480, 512, OFUSION, 0.008),
       "of512mu01": (640, 480, 512, OFUSION, 0.1), "sdf2048": (1280, 960, 2048, SDF, 0.1)}
KNOBS = ("SE_HIP_RAY_DEAL", "SE_HIP_PRIO", "SE_HIP_RAY_CACHE_LEVELS", "SE_HIP_NO_OVERLAP", "SE_HIP_DENSE", "SE_HIP_PRIO_SHARE")

def run(cfg, groups):
    W, H, N, field, mu = CFG[cfg]
    ref = None
    for g in groups:
        for k in KNOBS:
            os.environ.pop(k, None)
        if g != "-":
            for kv in g.split(";"):
                k, v = kv.split("=", 1)
                os.environ[k] = v
        s = SyntheticStream(W, H, 4.8)
        p = DenseSLAMPipeline((W, H), N, 4.8, field_type=field)
        for f in range(14):
            p.set_depth(s.depth(f)); p.setPose(s.pose(f))
            p.integration(s.k, 1, mu, f); p.raycasting(s.k, mu, f)
        p.sync()
        res = []
        for rep in range(3):
            p.enable_timing(True)
            for _ in range(20):
                p.raycasting(s.k, mu, 13)
            t = p.timings(reset=True)["raycast"]
            p.enable_timing(False)
            res.append(1e3 * t["ms_sum"] / t["launches"])
        v, n = p.vertex_normal()
        if ref is None:
            ref = (v.copy(), n.copy())
        same = np.array_equal(v.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(n.view(np.uint32), ref[1].view(np.uint32))
        print(f"{cfg} [{g}]: raycast " + " / ".join(f"{x:.1f}" for x in res) + f" us  identical {same}", flush=True)
        p.close()

if __name__ == "__main__":
    a = sys.argv[1:]
    i = a.index("--")
    for cfg in a[:i]:
        run(cfg, a[i + 1:] or ["-"])
