#!/usr/bin/env python3
"""What is the one 35-40 ms stall ~20 ms into sustained load: the GPU's clock ramp, or the host gate's
fallback path?  Runs the soak workload (320x240 stress stream -> 512^3, pipelined, no host sync) from an idle GPU and records
  * the host time at which every frame call returns (the host gate blocks inside the call, so a device stall shows up as one long call),
  * in a sampler thread at ~2 kHz: the shader clock (hwmon freq1_input / pp_dpm_sclk), socket power and gpu_busy_percent from sysfs,
and prints, for the longest frame call, the clock / power samples before, inside and after it.  Run twice: SE_HIP_HOST_GATE=1 (default)
and =0 (event-ordered, the host never waits).  usage: stall_probe.py [frames] [idle_seconds]"""
import glob
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from supereight_amd.pipeline import DenseSLAMPipeline, SDF  # noqa: E402
from supereight_amd.synthetic import StressStream, to_colmajor  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
W, H, N, PATH = 320, 240, 512, 360


def sysfs_sources():
    src = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        f = glob.glob(card + "/hwmon/hwmon*/freq1_input")
        if f:
            src["sclk_hz"] = f[0]
            pw = glob.glob(card + "/hwmon/hwmon*/power1_average") + glob.glob(card + "/hwmon/hwmon*/power1_input")
            if pw:
                src["power_uw"] = pw[0]
            if os.path.exists(card + "/gpu_busy_percent"):
                src["busy_pct"] = card + "/gpu_busy_percent"
            if os.path.exists(card + "/pp_dpm_sclk"):
                src["dpm_sclk"] = card + "/pp_dpm_sclk"
            break
    return src


SRC = sysfs_sources()
samples = []
stop = False


def read(path):
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return None


def sampler():
    while not stop:
        t = time.perf_counter()
        rec = [t]
        for key in ("sclk_hz", "power_uw", "busy_pct"):
            v = read(SRC[key]) if key in SRC else None
            rec.append(int(v) if v and v.lstrip("-").isdigit() else None)
        samples.append(rec)
        time.sleep(0.0004)


s = StressStream(W, H, 4.8)
depth = torch.from_numpy(np.stack([s.depth(f) for f in range(PATH)])).cuda()
poses = [to_colmajor(s.pose(f)) for f in range(PATH)]
k = np.ascontiguousarray(s.k, np.float32)
p = DenseSLAMPipeline((W, H), N, 4.8, field_type=SDF)
for f in range(8):                         # code paths warm, then let the GPU fall back to its idle state
    p.frame(depth[f].data_ptr(), poses[f], k, 0.1, f)
p.sync()
time.sleep(idle)
th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(0.02)
stamps = np.zeros(frames + 1)
stamps[0] = time.perf_counter()
for f in range(frames):
    p.frame(depth[(8 + f) % PATH].data_ptr(), poses[(8 + f) % PATH], k, 0.1, 8 + f)
    stamps[f + 1] = time.perf_counter()
p.sync()
t_end = time.perf_counter()
time.sleep(0.02)
stop = True
th.join()
p.close()

dt = np.diff(stamps)
worst = int(dt.argmax())
t0 = stamps[0]
smp = np.array([[r[0] - t0] + [np.nan if v is None else v for v in r[1:]] for r in samples], dtype=float)


def window(lo, hi):
    m = (smp[:, 0] >= lo) & (smp[:, 0] < hi)
    if not m.any():
        return None
    out = {"samples": int(m.sum())}
    for j, key in enumerate(("sclk_mhz", "power_w", "busy_pct"), start=1):
        col = smp[m, j]
        col = col[~np.isnan(col)]
        if col.size:
            scale = 1e-6 if key != "busy_pct" else 1.0
            out[key] = {"min": round(float(col.min() * scale), 1), "median": round(float(np.median(col) * scale), 1), "max": round(float(col.max() * scale), 1)}
    return out


a, b = stamps[worst] - t0, stamps[worst + 1] - t0
rep = {"workload": f"stress stream {W}x{H} -> {N}^3, {frames} pipelined frames from an idle GPU ({idle} s idle)",
       "host_gate": os.environ.get("SE_HIP_HOST_GATE", "1"), "sysfs": SRC, "sampler_hz": round(len(samples) / max(1e-9, samples[-1][0] - samples[0][0])) if len(samples) > 1 else 0,
       "total_ms": round(1e3 * (t_end - t0), 2), "fps_overall": round(frames / (t_end - t0), 1),
       "frame_call_us": {"median": round(1e6 * float(np.median(dt)), 1), "p99": round(1e6 * float(np.percentile(dt, 99)), 1), "max": round(1e6 * float(dt.max()), 1)},
       "longest_call": {"frame": worst, "starts_at_ms": round(1e3 * a, 2), "lasts_ms": round(1e3 * (b - a), 2)},
       "calls_over_1ms": [{"frame": int(i), "at_ms": round(1e3 * (stamps[i] - t0), 2), "ms": round(1e3 * float(dt[i]), 2)} for i in np.nonzero(dt > 1e-3)[0][:20]],
       "clock_before_stall": window(max(0.0, a - 0.010), a), "clock_inside_stall": window(a, b), "clock_after_stall": window(b, b + 0.010),
       "clock_first_5ms": window(0.0, 0.005), "clock_last_50ms": window((t_end - t0) - 0.05, t_end - t0)}
# coarse timeline: 2 ms bins of the first 120 ms (median sclk, frames completed by the host)
tl = []
for i in range(60):
    w = window(0.002 * i, 0.002 * (i + 1))
    tl.append({"ms": 2 * i, "sclk_mhz": (w or {}).get("sclk_mhz", {}).get("median"), "power_w": (w or {}).get("power_w", {}).get("median"),
               "frames_returned": int(((stamps[1:] - t0) < 0.002 * (i + 1)).sum())})
rep["timeline_2ms"] = tl
print(json.dumps(rep))
