"""Reads a rocprofv3 kernel trace CSV and prints the frame timeline of the last frames: kernel start / end per queue and the
gaps on the main queue (sweep -> raycast, raycast -> next sweep)."""
import csv
import glob
import statistics as S
import sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    k = "ray" if "k_raycast" in n else "sweep" if "k_integrate" in n else "scan" if "k_alloc_scan" in n else "fill" if "fillBuffer" in n else None
    if k:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id")))
ev.sort()
tail = ev[-200:]
t0 = tail[0][0]
for s, e, k, q in tail[-24:]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {k:6s} dur {(e - s) / 1e3:6.1f} q{q}")
main = [x for x in tail if x[2] in ("ray", "sweep")]
g1 = [(b[0] - a[1]) / 1e3 for a, b in zip(main, main[1:]) if a[2] == "sweep" and b[2] == "ray"]
g2 = [(b[0] - a[1]) / 1e3 for a, b in zip(main, main[1:]) if a[2] == "ray" and b[2] == "sweep"]
per = [(b[0] - a[0]) / 1e3 for a, b in zip([x for x in main if x[2] == "sweep"], [x for x in main if x[2] == "sweep"][1:])]
print("gap sweep->ray mean %.2f  ray->sweep mean %.2f  frame period mean %.2f" % (S.mean(g1), S.mean(g2), S.mean(per)))
scans = [x for x in tail if x[2] == "scan"]
sw = [x for x in tail if x[2] == "sweep"]
# scan end relative to the raycast end of the same period, and sweep start relative to scan end
rays = [x for x in tail if x[2] == "ray"]
d1 = []; d2 = []
for sc in scans:
    nxt = [x for x in sw if x[0] > sc[1]]
    prv = [x for x in rays if x[1] <= (nxt[0][0] if nxt else 1e30)]
    if nxt and prv:
        d1.append((nxt[0][0] - sc[1]) / 1e3); d2.append((nxt[0][0] - prv[-1][1]) / 1e3)
print("next sweep start - scan end: mean %.2f ; next sweep start - raycast end: mean %.2f" % (S.mean(d1), S.mean(d2)))
