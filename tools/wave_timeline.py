#!/usr/bin/env python3
"""Per-wave timeline of the fused raycast + scan launch (profiles/r05af_wave_timeline.md, r05ah_wave_timelines.txt).
Needs a probe build of the library -- the probe is not part of the product tree:
    git apply tools/wave_probe.patch && tools/build_variant.sh wlog && git apply -R tools/wave_probe.patch
    gpurun -- 'SE_HIP_LIB=$PWD/gpurun_ab/wlog.so SE_HIP_WLOG=/tmp/wlog.bin python tools/wave_timeline.py room sdf 512 0.1'
Lane 0 of every raycast / scan wave writes s_memrealtime (100 MHz) at entry and exit, HW_ID / XCC_ID and the wave maxima of first-leaf trips and march
batches into a pinned host buffer; the library dumps the buffer of the previous launch when the next one is enqueued.
usage: wave_timeline.py room|stress sdf|ofusion <volume resolution> <mu>"""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from supereight_amd.pipeline import DenseSLAMPipeline, SDF, OFUSION
from supereight_amd.synthetic import make_stream, to_colmajor
kind, field, N, mu, W, H = sys.argv[1], (SDF if sys.argv[2] == "sdf" else OFUSION), int(sys.argv[3]), float(sys.argv[4]), 640, 480
dim = 4.8
s = make_stream(kind, W, H, dim); F = 40
depth = torch.from_numpy(np.stack([s.depth(f) for f in range(F)])).cuda()
p = DenseSLAMPipeline((W, H), N, dim, field_type=field, streaming=True)
k = np.ascontiguousarray(s.k, np.float32)
for f in range(F):
    p.frame(depth[f].data_ptr(), to_colmajor(s.pose(f)), k, mu, f)
torch.cuda.synchronize()    # (no se_hip call: the file holds the log of the last-but-one fused launch, dumped when the last one was enqueued)
raw = np.fromfile(os.environ["SE_HIP_WLOG"], dtype=np.uint64).reshape(-1, 4)
a = raw[:16384]; a = a[a[:, 1] > 0]
sc = raw[16384:]; sc = sc[sc[:, 1] > 0]
t0 = a[:, 0].astype(np.int64); t1 = a[:, 1].astype(np.int64)
base = t0.min(); st = (t0 - base) / 100.0; en = (t1 - base) / 100.0
hw = a[:, 2] & np.uint64(0xFFFFFFFF); xcc = (a[:, 2] >> np.uint64(32)) & np.uint64(0xF)
key = ((xcc.astype(np.int64) * 8 + ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)) * 2 + ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64))
key = (key * 16 + ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)) * 4 + ((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64)
batches = ((a[:, 3] >> np.uint64(10)) & np.uint64(0x3FF)).astype(np.int64); trips = ((a[:, 3] >> np.uint64(20)) & np.uint64(0x3FF)).astype(np.int64)
dur = en - st
print("==", kind, sys.argv[2], N, "mu", mu)
print("waves", len(a), "span_us", round(en.max(), 2), "start max", round(st.max(), 2), "dur mean", round(dur.mean(), 2), "p50", round(np.median(dur), 2), "p90", round(np.percentile(dur, 90), 2), "p99", round(np.percentile(dur, 99), 2), "max", round(dur.max(), 2))
uk, inv = np.unique(key, return_inverse=True)
fin = np.array([en[inv == j].max() for j in range(len(uk))])
print("simds", len(uk), "finish p10/p50/p90/max", round(np.percentile(fin, 10), 1), round(np.median(fin), 1), round(np.percentile(fin, 90), 1), round(fin.max(), 1))
ts = np.linspace(0, en.max(), 12)
print("resident waves at t:", [(round(float(t), 1), int(((st <= t) & (en > t)).sum())) for t in ts])
A = np.vstack([trips, batches, np.ones(len(dur))]).T; co = np.linalg.lstsq(A, dur, rcond=None)[0]
print("dur ~ %.3f us * max_trips + %.3f us * max_batches + %.2f" % tuple(co), "corr", round(float(np.corrcoef(A @ co, dur)[0, 1]), 3))
print("trips mean/max", round(trips.mean(), 1), trips.max(), "batches mean/max", round(batches.mean(), 1), batches.max())
o = np.argsort(-en)[:8]
print("last waves (end, trips, batches):", [(round(float(en[i]), 1), int(trips[i]), int(batches[i])) for i in o])
half = en.max() * 0.6
print("waves alive after 60%% of the span: %d (%.1f %%)" % ((en > half).sum(), 100.0 * (en > half).mean()))

if len(sc):
    s0 = (sc[:, 0].astype(np.int64) - base) / 100.0; s1 = (sc[:, 1].astype(np.int64) - base) / 100.0
    print("scan waves", len(sc), "start p10/p50/p90/max", round(np.percentile(s0, 10), 1), round(np.median(s0), 1), round(np.percentile(s0, 90), 1), round(s0.max(), 1),
          "end p50/p90/max", round(np.median(s1), 1), round(np.percentile(s1, 90), 1), round(s1.max(), 1), "dur mean/max", round((s1 - s0).mean(), 1), round((s1 - s0).max(), 1))
    ts = np.linspace(0, max(en.max(), s1.max()), 12)
    print("resident scan waves at t:", [(round(float(t), 1), int(((s0 <= t) & (s1 > t)).sum())) for t in ts])
