#!/usr/bin/env python3
"""Check bench.py's roofline duration against a rocprofv3 kernel trace: the ten mfa_jit_kernel dispatches of a step
overlap (several streams), so the device time of a step is first start -> last end of its ten dispatches.
usage: span_from_trace.py <..._kernel_trace.csv> [dispatches per step = 10] [set-up passes = 7] [warm-up steps = 1] [timed steps = 3]
(set-up passes of the default bench run: one back-to-back calibration pass + two passes for each of three stream counts)"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("mfa_jit_kernel")]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n_setup = int(sys.argv[3]) if len(sys.argv) > 3 else 7
n_warm = int(sys.argv[4]) if len(sys.argv) > 4 else 1
n_timed = int(sys.argv[5]) if len(sys.argv) > 5 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
labels = ["set-up pass %d%s" % (k + 1, " (back to back, one stream)" if k == 0 else "") for k in range(n_setup)]
labels += ["warm-up step %d" % (k + 1) for k in range(n_warm)] + ["timed step %d" % (k + 1) for k in range(n_timed)]
print("mfa_jit_kernel dispatches: %d (later ones belong to the secondary lines)" % len(rows))
for g, label in enumerate(labels):
    grp = rows[g * per:(g + 1) * per]
    if len(grp) < per:
        break
    t0 = min(int(r["Start_Timestamp"]) for r in grp)
    t1 = max(int(r["End_Timestamp"]) for r in grp)
    tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp)
    print("%s: span %.3f ms, sum of the %d kernel durations %.3f ms" % (label, (t1 - t0) / 1e6, per, tot / 1e6))
