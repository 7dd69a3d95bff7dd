"""Diagnostic: raycast kernel time vs number of image rows cast (latency- or throughput-bound?)."""
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
for rows in ("0,480", "0,240", "120,360", "0,120", "240,360", "0,64", "200,264", "200,232", "200,216", "200,208"):
    os.environ["SE_HIP_DEBUG_RAY_ROWS"] = rows
    p.enable_timing(True)
    for _ in range(20):
        p.raycasting(s.k, mu, 13)
    t = p.timings(reset=True)["raycast"]
    p.enable_timing(False)
    b, e = map(int, rows.split(","))
    print(f"rows {rows:>8}: {1e3 * t['ms_sum'] / t['launches']:7.1f} us  ({(e - b) * W // 64} waves)")
