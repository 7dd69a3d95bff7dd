"""Diagnostic: raycast kernel time with phases disabled (full image, full load)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supereight_amd.pipeline import DenseSLAMPipeline
from supereight_amd.synthetic import SyntheticStream
W, H, N, dim, mu = 640, 480, 512, 4.8, 0.1
s = SyntheticStream(W, H, dim)
p = DenseSLAMPipeline((W, H), N, dim)
for f in range(14):
    p.set_depth(s.depth(f)); p.setPose(s.pose(f))
    p.integration(s.k, 1, mu, f)
p.sync()
for ph, name in ((0, "full"), (2, "no gradient"), (1, "first-leaf search only"), (5, "ray set-up + LDS staging + stores only")):
    os.environ["SE_HIP_DEBUG_RAY_PHASES"] = str(ph)
    p.enable_timing(True)
    for _ in range(20):
        p.raycasting(s.k, mu, 13)
    t = p.timings(reset=True)["raycast"]
    p.enable_timing(False)
    print(f"{name:>24}: {1e3 * t['ms_sum'] / t['launches']:7.1f} us")
