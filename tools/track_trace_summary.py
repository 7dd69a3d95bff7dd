"""Kernel trace of the tracked loop (rocprofv3 --kernel-trace over tools/track_probe.py --trace N) -> markdown: one frame's launches in
order (duration, idle gap in front of each) and per-kernel averages over the last frames.
  python tools/track_trace_summary.py <dir or kernel_trace.csv> [frames]"""
import csv
import glob
import os
import sys


def main():
    src = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    if os.path.isdir(src):
        src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
    ends = [i for i, r in enumerate(rows) if name(r).startswith("k_icp_finish")]
    if len(ends) < 3:
        print("no tracked frames in the trace"); return
    # a frame = (previous k_icp_finish*, this one], shifted so that it starts at the frame's first kernel after the raycast
    starts = [i for i, r in enumerate(rows) if name(r).startswith("k_depth_pyramid") or name(r).startswith("k_bilateral")]
    if len(starts) < 3:   # traces older than k_depth_pyramid: the frame's first launch was the runtime's copy of the depth image
        starts = [i for i, r in enumerate(rows) if "copyBuffer" in name(r)]
    frames = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][-last:]
    print(f"# tracked loop, kernel trace ({os.path.basename(src)}; the last {len(frames)} frames)\n")
    a, b = frames[len(frames) // 2]
    print("## one frame (launch order; `gap` = idle time of the queue in front of the launch)\n")
    print("| kernel | dur us | gap us |\n|---|---|---|")
    prev = int(rows[a - 1]["End_Timestamp"]) if a > 0 else None
    tot_k = tot_g = 0.0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        g = max(0.0, (s - prev) / 1e3) if prev is not None else 0.0
        print(f"| {name(r)[:48]} | {(e - s) / 1e3:.2f} | {g:.2f} |")
        tot_k += (e - s) / 1e3; tot_g += g
        prev = e
    print(f"\nframe: {tot_k:.1f} us in kernels + {tot_g:.1f} us of gaps = {tot_k + tot_g:.1f} us\n")
    agg = {}
    nfr = 0
    span = 0.0
    for a, b in frames:
        nfr += 1
        span += (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
        for r in rows[a:b]:
            d = agg.setdefault(name(r), [0, 0.0])
            d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"## per frame, averaged over {nfr} frames (frame period {span / nfr:.1f} us under the tracer)\n")
    print("| kernel | launches / frame | us / frame | avg us |\n|---|---|---|---|")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k[:60]} | {n / nfr:.1f} | {t / nfr:.1f} | {t / n:.2f} |")
    print(f"| total | | {sum(t for _, t in agg.values()) / nfr:.1f} | |")


if __name__ == "__main__":
    main()
