#!/bin/bash
R=$PWD; out=$R/gpurun_out/r04_c3lu; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --output-format csv --kernel-trace -d $out/tr -o t -- python3 $R/tools/config3lu_line.py > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
tail -1 $out/run.log | cut -c1-600
python3 - <<PY
import csv, glob
f = glob.glob("$out/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("region_scan", "walk_kernel", "walk_lean", "mfa_jit"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-40:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%9.3f ms  +%7.3f ms  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"].split("(")[0][:60]))
PY
