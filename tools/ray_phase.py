"""Diagnostic: per-wave phase clocks of the raycast kernel on the bench stream (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from supereight_amd.pipeline import DenseSLAMPipeline, SDF, OFUSION
from supereight_amd.synthetic import SyntheticStream
W, H, N, dim, mu = 640, 480, int(os.environ.get("RES", 512)), 4.8, float(os.environ.get("MU", 0.1))
FIELD = OFUSION if os.environ.get("FIELD", "sdf") == "ofusion" else SDF
s = SyntheticStream(W, H, dim)
p = DenseSLAMPipeline((W, H), N, dim, field_type=FIELD)
for f in range(14):
    p.set_depth(s.depth(f)); p.setPose(s.pose(f))
    p.integration(s.k, 1, mu, f)
p.sync()
for rows in ("0,480", "200,208"):
    os.environ["SE_HIP_DEBUG_RAY_ROWS"] = rows
    b, e = map(int, rows.split(","))
    waves = (W // 8) * ((e - b) // 8)
    for stats in (False, True):
        p.enable_stats(stats)
        p.enable_timing(True)
        for _ in range(10):
            p.raycasting(s.k, mu, 13)
        t = p.timings(reset=True)["raycast"]
        p.enable_timing(False)
        print(f"rows {rows} stats={stats}: kernel {1e3 * t['ms_sum'] / t['launches']:.1f} us")
    st = p.stats()
    n = 10 * waves
    print("  per-wave avg cycles: stage %.0f iter %.0f march %.0f grad %.0f | max wave %d | gets/wave %.1f interps/wave %.1f" % (
        st["clk_stage"] / n, st["clk_iter"] / n, st["clk_march"] / n, st["clk_grad"] / n, st["clk_wave_max"], st["gets"] / n, st["interps"] / n))
    print("  max over waves: iter %d march %d grad %d" % (st["r13"], st["r14"], st["r15"]))
    p.enable_stats(False)
