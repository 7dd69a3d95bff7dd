"""Diagnostic: integration sweep kernel time with the arithmetic or the voxel traffic removed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supereight_amd.pipeline import DenseSLAMPipeline
from supereight_amd.synthetic import SyntheticStream
W, H, N, dim, mu = 640, 480, 512, 4.8, 0.1
os.environ["SE_HIP_NO_OVERLAP"] = "1"
s = SyntheticStream(W, H, dim)
p = DenseSLAMPipeline((W, H), N, dim)
for f in range(14):
    p.set_depth(s.depth(f)); p.setPose(s.pose(f))
    p.integration(s.k, 1, mu, f)
p.sync()
for mode, name in ((0, "full"), (1, "voxel copy only (no arithmetic)"), (2, "arithmetic only (no voxel traffic)"), (0, "full")):
    os.environ["SE_HIP_DEBUG_INTEG"] = str(mode)
    p.enable_timing(True)
    for _ in range(20):
        p.integrate_sweep(s.k, 1, mu, 13)
    t = p.timings(reset=True)["integrate"]
    p.enable_timing(False)
    print(f"{name:>36}: {1e3 * t['ms_sum'] / t['launches']:7.1f} us")
