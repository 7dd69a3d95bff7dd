"""Diagnostic (GPU): raycast kernel time for the workgroup -> tile mappings (SE_HIP_XCD_SWIZZLE = 0 / 2 / 4 / 8)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from supereight_amd.pipeline import DenseSLAMPipeline, SDF, OFUSION
from supereight_amd.synthetic import SyntheticStream

def run(W, H, N, field, mu, modes):
    ref = None
    for mode in modes:
        s = SyntheticStream(W, H, 4.8)
        os.environ["SE_HIP_XCD_SWIZZLE"] = str(mode)
        p = DenseSLAMPipeline((W, H), N, 4.8, field_type=field)
        for f in range(14):
            p.set_depth(s.depth(f)); p.setPose(s.pose(f))
            p.integration(s.k, 1, mu, f); p.raycasting(s.k, mu, f)
        p.sync()
        p.enable_timing(True)
        for _ in range(30):
            p.raycasting(s.k, mu, 13)
        t = p.timings(reset=True)["raycast"]
        p.enable_timing(False)
        v, n = p.vertex_normal()
        if ref is None:
            ref = (v.copy(), n.copy())
        same = np.array_equal(v.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(n.view(np.uint32), ref[1].view(np.uint32))
        print(f"{W}x{H} -> {N}^3 field {field} swizzle {mode}: raycast {1e3 * t['ms_sum'] / t['launches']:7.1f} us  identical-to-mode-0 {same}", flush=True)
        p.close()

if __name__ == "__main__":
    which = sys.argv[1:] or ["sdf512", "sdf1024", "of512"]
    if "sdf512" in which: run(640, 480, 512, SDF, 0.1, (0, 2, 4, 8, 16))
    if "sdf1024" in which: run(640, 480, 1024, SDF, 0.1, (0, 4, 8))
    if "of512" in which: run(640, 480, 512, OFUSION, 0.1, (0, 4, 8))
    if "sdf2048" in which: run(1280, 960, 2048, SDF, 0.1, (0, 4, 8))
