#!/usr/bin/env python3
"""Check bench.py's roofline duration against a rocprofv3 kernel trace: the ten mfa_jit_kernel dispatches of a step
overlap (ten streams), so the device time of a step is first start -> last end of its ten dispatches.
usage: span_from_trace.py <..._kernel_trace.csv> [dispatches per step = 10]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("mfa_jit_kernel")]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_head = min(len(rows), 50)            # calibration pass (back to back) + warm-up + 3 steps of the headline; later dispatches belong to the secondary lines
print("mfa_jit_kernel dispatches: %d" % len(rows))
for s in range(0, n_head, per):
    grp = rows[s:s + per]
    t0 = min(int(r["Start_Timestamp"]) for r in grp)
    t1 = max(int(r["End_Timestamp"]) for r in grp)
    tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp)
    print("%s: span %.3f ms, sum of the %d kernel durations %.3f ms" % (["calibration pass (one stream)", "warm-up step", "timed step 1", "timed step 2", "timed step 3"][s // per], (t1 - t0) / 1e6, per, tot / 1e6))
