"""Diagnostic (GPU, -DSE_DIAG library: tools/build_variant.sh diag -DSE_DIAG; SE_HIP_LIB=gpurun_ab/diag.so):
per-pixel iterator trips / march batches and per-wave clocks of one raycast launch -> gpurun_out/ray_diag_<tag>.npz."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from supereight_amd.pipeline import DenseSLAMPipeline, SDF, OFUSION, load_library
from supereight_amd.synthetic import SyntheticStream

W, H, N, dim, mu = 640, 480, int(os.environ.get("RES", 512)), 4.8, float(os.environ.get("MU", 0.1))
FIELD = OFUSION if os.environ.get("FIELD", "sdf") == "ofusion" else SDF
tag = os.environ.get("TAG", "sdf512")
lib = load_library()
lib.se_hip_diag_enable.argtypes = [C.c_void_p, C.c_int32]
lib.se_hip_diag_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
s = SyntheticStream(W, H, dim)
p = DenseSLAMPipeline((W, H), N, dim, field_type=FIELD)
for f in range(14):
    p.set_depth(s.depth(f)); p.setPose(s.pose(f))
    p.integration(s.k, 1, mu, f)
    p.raycasting(s.k, mu, f)
p.sync()
p.enable_stats(True)
for _ in range(3):
    p.raycasting(s.k, mu, 13)
p.sync()
assert lib.se_hip_diag_enable(p._h, 1) == 0
p.enable_timing(True)
p.raycasting(s.k, mu, 13)
p.sync()
t = p.timings(reset=True)["raycast"]
print(f"diag launch: {1e3 * t['ms_sum'] / t['launches']:.1f} us by events")
p.enable_timing(False)
pix = np.zeros((H, W), np.uint32)
nw = (W // 8) * (H // 8)
wave = np.zeros((nw, 8), np.uint32)
assert lib.se_hip_diag_download(p._h, pix.ctypes.data, wave.ctypes.data) == 0
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(f"gpurun_out/ray_diag_{tag}.npz", pix=pix, wave=wave)
trips, batches = pix & 0xFFFF, pix >> 16
t0 = wave[:, 0].astype(np.int64); t1 = wave[:, 1].astype(np.int64)   # s_memrealtime, 100 MHz
start = ((t0 - t0[0] + 2**31) % 2**32) - 2**31
dur = (t1 - t0) % 2**32
start -= start.min()
end = start + dur
q = lambda a: np.percentile(a, [0, 10, 50, 90, 99, 100]).round(1)
print("microseconds: wave start p0/10/50/90/99/100:", q(start / 100.0))
print("wave dur    :", q(dur / 100.0))
print("wave end    :", q(end / 100.0))
cyc = (wave[:, 2].astype(np.int64) + wave[:, 3] + wave[:, 4] + wave[:, 5])
print("shader cycles per wave (stage+iter+march+grad):", q(cyc), " -> clock %.2f GHz" % (cyc.sum() / dur.sum() / 10.0))
print("stage       :", q(wave[:, 2])); print("iter        :", q(wave[:, 3])); print("march       :", q(wave[:, 4])); print("grad+store  :", q(wave[:, 5]))
print("wave trips  :", q(wave[:, 6] & 0xFFFF), " wave batches:", q(wave[:, 6] >> 16))
print("lane trips  :", q(trips), " lane batches:", q(batches), " mean trips %.1f batches %.2f" % (trips.mean(), batches.mean()))
order = np.argsort(-end)[:16]
print("16 last-finishing waves: (tile_x, tile_y, start us, end us, iter, march, trips, batches)")
for i in order:
    print("  ", i % (W // 8), i // (W // 8), start[i] / 100.0, end[i] / 100.0, wave[i, 3], wave[i, 4], wave[i, 6] & 0xFFFF, wave[i, 6] >> 16)
T = end.max()
print("resident waves over time:", [int(((start <= f * T) & (end > f * T)).sum()) for f in (0.02, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95)], "T = %.1f us" % (T / 100.0))
hw = wave[:, 7]
xcc = (hw >> 28).astype(np.int64); hid = (hw & 0x0FFFFFFF).astype(np.int64)
key = (xcc << 20) | (((hid >> 13) & 7) << 12) | (((hid >> 12) & 1) << 8) | (((hid >> 8) & 15) << 4) | ((hid >> 4) & 3)   # one value per SIMD
u, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
fin = np.zeros(len(u)); np.maximum.at(fin, inv, end / 100.0)
cost = (wave[:, 6] & 0xFFFF).astype(np.float64) + 5 * (wave[:, 6] >> 16)
csum = np.zeros(len(u)); np.add.at(csum, inv, cost)
print("SIMDs", len(u), "waves per SIMD histogram:", np.bincount(cnt).tolist())
print("SIMD finish time us p0/10/50/90/99/100:", q(fin))
print("SIMD cost sum (trips + 5 * batches):", q(csum), " corr(finish, cost sum) = %.2f" % np.corrcoef(fin, csum)[0, 1])
p.close()
