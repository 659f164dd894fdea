#!/usr/bin/env python3
"""Start/end offsets (ms) of the mfa_jit_kernel dispatches of the LAST headline step in a rocprofv3 kernel trace.
usage: timeline.py <..._kernel_trace.csv> [dispatches per step = 10]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("mfa_jit_kernel")]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
grp = rows[-per:]
t0 = min(int(r["Start_Timestamp"]) for r in grp)
for r in grp:
    print("grid %6s lds %6s vgpr %4s  start %7.3f  end %7.3f  dur %7.3f" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"),
          (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
