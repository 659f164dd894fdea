#!/usr/bin/env python3
"""Check bench.py's roofline duration against a rocprofv3 kernel trace.  A step = R region_scan_kernel dispatches on the region
stream and W walk dispatches (walk_kernel, or mfa_jit_kernel) on the walk streams, overlapping: its device time is first start -> last end of those
R + W dispatches.  Also prints the average duration of each kernel over the timed steps (what `--stats` averages, restricted to them).
usage: span_from_trace.py <..._kernel_trace.csv> [region launches per step = 3] [walk launches per step = 10] [set-up passes = 1]
       [warm-up steps = 1] [timed steps = 3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n_setup = int(sys.argv[4]) if len(sys.argv) > 4 else 1
n_warm = int(sys.argv[5]) if len(sys.argv) > 5 else 1
n_timed = int(sys.argv[6]) if len(sys.argv) > 6 else 3
region = sorted((r for r in rows if "region_scan_kernel" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
WALK = ("walk_kernel", "mfa_jit_kernel")      # the table-driven walk, or the kernels generated per automaton
walk = sorted((r for r in rows if any(w in r["Kernel_Name"] for w in WALK)), key=lambda r: int(r["Start_Timestamp"]))
labels = ["set-up pass %d" % (k + 1) for k in range(n_setup)]
labels += ["warm-up step %d" % (k + 1) for k in range(n_warm)] + ["timed step %d" % (k + 1) for k in range(n_timed)]
print("region_scan_kernel dispatches: %d, walk dispatches: %d (later ones belong to the secondary lines)" % (len(region), len(walk)))
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tr, tw = [], []
for g, label in enumerate(labels):
    gr, gw = region[g * R:(g + 1) * R], walk[g * W:(g + 1) * W]
    if len(gr) < R or len(gw) < W:
        break
    t0 = min(int(r["Start_Timestamp"]) for r in gr + gw)
    t1 = max(int(r["End_Timestamp"]) for r in gr + gw)
    print("%s: span %.3f ms; region_scan_kernel %d launches, sum %.3f ms; walk kernels %d launches, sum %.3f ms" % (
        label, (t1 - t0) / 1e6, R, sum(map(dur, gr)) / 1e6, W, sum(map(dur, gw)) / 1e6))
    if label.startswith("timed"):
        tr += [dur(r) for r in gr]; tw += [dur(r) for r in gw]
if tr:
    print("timed steps: region_scan_kernel average %.3f ms per launch; walk kernels average %.3f ms per launch" % (
        sum(tr) / len(tr) / 1e6, sum(tw) / len(tw) / 1e6))
