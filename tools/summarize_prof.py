#!/usr/bin/env python3
"""Condenses rocprofv3 output (kernel-trace --stats CSV + PMC counter CSVs of separate passes) into a
markdown summary: per-kernel launch count / average duration, and per-kernel per-launch counter means."""
import csv
import glob
import os
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]
OURS = ("k_alloc", "k_integrate", "k_raycast", "k_fill", "k_occ_commit", "k_zero_chain", "k_mm2meters", "k_icp", "k_half_sample", "k_depth2vertex", "k_vertex2normal", "k_bilateral")


def short(name):
    for o in OURS:
        if o in name:
            i = name.index(o)
            j = name.find("(", i)
            return name[i:j] if j > 0 else name[i:]
    return name[:60]


def find(sub, pattern):
    return sorted(glob.glob(os.path.join(out_dir, sub, "**", pattern), recursive=True))


LAST = int(os.environ.get("SE_PROF_LAST", 50))   # launches per kernel that belong to the timed region (= bench.py --steps; the prewarm / warm-up launches come first)
print(f"# rocprofv3 summary `{tag}`  (bench.py --steps {LAST} --warmup 10 --no-modes --sustain 0, MI355X; per-kernel figures = the last {LAST} launches)\n")
stats = find("trace", "*kernel_stats.csv")
if stats:
    print("## kernel trace (--kernel-trace --stats)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    with open(stats[0]) as fh:
        for row in csv.DictReader(fh):
            n = row.get("Name", "")
            tot = float(row.get("TotalDurationNs", 0)) / 1e6
            print(f"| {short(n)} | {row.get('Calls')} | {tot:.3f} | {float(row.get('AverageNs', 0)) / 1e3:.2f} | "
                  f"{float(row.get('MinNs', 0)) / 1e3:.2f} | {float(row.get('MaxNs', 0)) / 1e3:.2f} | {row.get('Percentage')} |")
    print()
# steady-state durations from the raw trace: skip each kernel's first 14 launches (warm-up frames)
trace = find("trace", "*kernel_trace.csv")
if trace:
    dur = defaultdict(list)
    with open(trace[0]) as fh:
        for row in csv.DictReader(fh):
            dur[short(row["Kernel_Name"])].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    print(f"## timed region (the last {LAST} launches of each kernel, from the raw kernel trace)\n")
    print("| kernel | launches | avg us | p50 us | p95 us |")
    print("|---|---|---|---|---|")
    for k, v in sorted(dur.items()):
        v.sort()
        d = sorted(x[1] for x in v[-LAST:])
        if not any(o in k for o in OURS):
            continue
        print(f"| {k} | {len(d)} | {sum(d) / len(d) / 1e3:.2f} | {d[len(d) // 2] / 1e3:.2f} | {d[int(len(d) * 0.95)] / 1e3:.2f} |")
    print()
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_cache", "pmc_tlb", "pmc_lat"):
    files = find(sub, "*counter_collection.csv")
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    with open(files[0]) as fh:
        for row in csv.DictReader(fh):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(f"## {sub}: mean counter value per launch (steady-state launches)\n")
    names = sorted({c for k in acc for c in acc[k]})
    print("| kernel | launches | " + " | ".join(names) + " |")
    print("|---|---|" + "---|" * len(names))
    for k in sorted(acc):
        if not any(o in k for o in OURS):
            continue
        cells = []
        n = 0
        for c in names:
            v = acc[k].get(c, [])
            v = v[-LAST:]
            n = max(n, len(v))
            cells.append(f"{sum(v) / len(v):.4g}" if v else "-")
        print(f"| {k} | {n} | " + " | ".join(cells) + " |")
    print()

# HBM traffic per launch = 2 * FETCH_SIZE (gfx950: the counter reports half of the bytes of a coalesced
# read, MI355X_MICROARCH.md section HBM; confirmed here on k_integrate, whose 2*FETCH + WRITE matches its
# algorithmic bytes) + WRITE_SIZE (calibrated on k_fill: 1.05 GB reported for 1.074 GB stored); counters are KB.
import json
fetch, write = {}, {}
nlaunch = {}
last_seen = {}
for sub, dst in (("pmc_fetch", fetch), ("pmc_write", write)):
    files = find(sub, "*counter_collection.csv")
    if not files:
        continue
    acc = defaultdict(list)
    with open(files[0]) as fh:
        for n_row, row in enumerate(csv.DictReader(fh)):
            if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
                last_seen[short(row["Kernel_Name"])] = n_row
    for k, v in acc.items():
        nlaunch[k] = max(nlaunch.get(k, 0), len(v))
        v = v[-LAST:]
        dst[k] = sum(v) / len(v)
names = {"k_alloc_scan": "alloc_scan", "k_integrate": "integrate", "k_raycast": "raycast"}
kern = {}
# several kernels can share a prefix (r04: k_raycast_scan = raycast + the next frame's scan in one launch, beside a few stand-alone
# k_raycast launches of the warm-up frames): the one with the most launches is the frame loop's
# ... or rather: the one still being launched when the run ends (rows are in dispatch order; the timed loop comes last, the pre-warm's hundreds of
# stand-alone k_raycast launches first), provided it has the timed region's worth of launches
# (bench.py's closed-loop leg runs behind the timed loop and launches stand-alone k_raycast again: where the fused k_raycast_scan has the timed
# region's launches it IS the timed loop's raycast launch, whatever comes later; tools/gpu_profile.sh now passes --no-closed-loop)
for k in sorted(fetch, key=lambda kk: (nlaunch.get(kk, 0) >= LAST, kk.startswith("k_raycast_scan"), last_seen.get(kk, 0))):
    for pre, nice in names.items():
        if k.startswith(pre) and k in write:
            kern[nice] = {"FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write[k],
                          "traffic_bytes": (2.0 * fetch[k] + write[k]) * 1024.0, "kernel": k}
wl = {"width": int(os.environ.get("SE_PROF_W", 640)), "height": int(os.environ.get("SE_PROF_H", 480)),
      "res": int(os.environ.get("SE_PROF_RES", 512)), "field": os.environ.get("SE_PROF_FIELD", "sdf"), "mu": float(os.environ.get("SE_PROF_MU", 0.1))}
with open(os.path.join(out_dir, "pmc_traffic.json"), "w") as fh:
    json.dump({"workload": wl, "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE) KB", "kernels": kern}, fh, indent=1)
