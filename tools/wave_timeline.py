#!/usr/bin/env python3
"""Per-wave timeline of a raycast launch (profiles/r05af_wave_timeline.md, r06*_wave_timeline_*.txt).  Needs a probe build of the library:
    tools/build_variant.sh wlog -DSE_WAVE_PROBE
    gpurun -- 'SE_HIP_LIB=$PWD/gpurun_ab/wlog.so SE_HIP_WLOG=/tmp/wlog.bin python tools/wave_timeline.py sdf1024 [--closed]'
Lane 0 of every raycast wave writes s_memrealtime (100 MHz) at entry / after the first-leaf search / at exit, HW_ID / XCC_ID and the wave maxima of
first-leaf trips and march batches into a pinned host buffer; the library dumps the buffer of the previous launch when the next one is prepared.
usage: wave_timeline.py <cfg of lib_ab.py> [--closed]     (--closed: one sync per frame, i.e. the stand-alone k_raycast behind its frame's sweep; default:
the streaming loop's fused launch)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from lib_ab import CFG, frames_of
from supereight_amd.pipeline import DenseSLAMPipeline, OFUSION, SDF
from supereight_amd.synthetic import to_colmajor

cfg = sys.argv[1]
closed = "--closed" in sys.argv
W, H, N, field, mu, n = CFG[cfg]
n = min(n, 40)
depth, poses, k = frames_of(cfg, n)
dev = torch.from_numpy(depth).cuda()
k32 = np.ascontiguousarray(k, np.float32)
kw = {}
if cfg.startswith("pooled"):
    kw["max_blocks"] = {512: 1 << 16, 1024: 1 << 19}.get(N, 1 << 21)
p = DenseSLAMPipeline((W, H), N, 4.8, field_type=SDF if field == "sdf" else OFUSION, streaming=not closed, **kw)
for f in range(n):
    p.frame(dev[f].data_ptr(), to_colmajor(poses[f]), k32, mu, f)
    if closed:
        p.sync()
torch.cuda.synchronize()    # (no se_hip call: the file holds the log of the last-but-one raycast launch, dumped when the last one was prepared)
raw = np.fromfile(os.environ["SE_HIP_WLOG"], dtype=np.uint64).reshape(-1, 4)
a = raw[raw[:, 1] > 0]
t0 = a[:, 0].astype(np.int64); t1 = a[:, 1].astype(np.int64)
base = t0.min(); st = (t0 - base) / 100.0; en = (t1 - base) / 100.0
leaf = ((a[:, 3] >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.int64) / 100.0     # first-leaf phase (incl. set-up, staging, beam start)
hw = a[:, 2] & np.uint64(0xFFFFFFFF); xcc = (a[:, 2] >> np.uint64(32)) & np.uint64(0xF)
key = ((xcc.astype(np.int64) * 8 + ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)) * 2 + ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64))
key = (key * 16 + ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)) * 4 + ((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64)
batches = ((a[:, 3] >> np.uint64(10)) & np.uint64(0x3FF)).astype(np.int64); trips = ((a[:, 3] >> np.uint64(20)) & np.uint64(0x3FF)).astype(np.int64)
dur = en - st
q = lambda v, pc: round(float(np.percentile(v, pc)), 2)
print("==", cfg, "closed loop (stand-alone k_raycast)" if closed else "streaming (fused launch, raycast waves)")
print("waves", len(a), "span_us", round(en.max(), 2), "start max", round(st.max(), 2), "dur mean", round(dur.mean(), 2), "p50", q(dur, 50), "p90", q(dur, 90), "p99", q(dur, 99), "max", round(dur.max(), 2))
print("first-leaf phase (set-up + beam start + search) us: mean", round(leaf.mean(), 2), "p50", q(leaf, 50), "p90", q(leaf, 90), "p99", q(leaf, 99), "max", round(leaf.max(), 2),
      "| march + gradient: mean", round((dur - leaf).mean(), 2), "p50", q(dur - leaf, 50), "p90", q(dur - leaf, 90), "p99", q(dur - leaf, 99), "max", round((dur - leaf).max(), 2))
uk, inv = np.unique(key, return_inverse=True)
fin = np.array([en[inv == j].max() for j in range(len(uk))])
cnt = np.bincount(inv)
print("simds", len(uk), "waves per simd min/max", cnt.min(), cnt.max(), "finish p10/p50/p90/max", q(fin, 10), q(fin, 50), q(fin, 90), round(fin.max(), 1))
ts = np.linspace(0, en.max(), 12)
print("resident waves at t:", [(round(float(t), 1), int(((st <= t) & (en > t)).sum())) for t in ts])
A = np.vstack([trips, batches, np.ones(len(dur))]).T; co = np.linalg.lstsq(A, dur, rcond=None)[0]
print("dur ~ %.3f us x trips + %.3f us x batches + %.2f us   (r = %.2f)" % (co[0], co[1], co[2], np.corrcoef(A @ co, dur)[0, 1]))
print("trips per wave (max over lanes): mean", round(trips.mean(), 1), "p90", q(trips, 90), "max", trips.max(), "| batches: mean", round(batches.mean(), 1), "p90", q(batches, 90), "max", batches.max())
late = en > np.percentile(en, 94)
print("the", int(late.sum()), "waves that end last (after %.1f us): trips mean %.1f, batches mean %.1f, first-leaf phase mean %.1f us, march mean %.1f us" %
      (np.percentile(en, 94), trips[late].mean(), batches[late].mean(), leaf[late].mean(), (dur - leaf)[late].mean()))
order = np.argsort(-en)[:12]
print("the twelve last waves (end, trips, batches, leaf us, march us):", [(round(float(en[i]), 1), int(trips[i]), int(batches[i]), round(float(leaf[i]), 1), round(float(dur[i] - leaf[i]), 1)) for i in order])
# per-XCD view
for x in sorted(set(xcc.tolist())):
    mk = xcc == x
    print("  xcc", int(x), "waves", int(mk.sum()), "finish", round(float(en[mk].max()), 1), "mean dur", round(float(dur[mk].mean()), 1))
p.close()
