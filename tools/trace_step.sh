#!/bin/bash
# usage: tools/trace_step.sh <tag> [env...]  -- kernel trace of a short headline run; prints the kernels of the LAST step with start offsets and durations
tag=$1; shift
R=$PWD; out=$R/gpurun_out/$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace -d $out/tr -o t -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 3 --warmup 3 > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$out/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("region_scan", "walk_kernel", "gate_wait"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step = from the last-but-zero'th region launch group: find starts of steps as region launches that follow a walk
regs = [i for i, r in enumerate(rows) if "region_scan" in r["Kernel_Name"]]
# steps are separated by gaps: take the rows after the last gap > 1 ms between consecutive starts? simpler: last N rows beginning at the last region launch that starts after every earlier kernel ended
ends = 0; start_idx = 0
for i, r in enumerate(rows):
    s = int(r["Start_Timestamp"])
    if "region_scan" in r["Kernel_Name"] and s >= ends: start_idx = i
    ends = max(ends, int(r["End_Timestamp"]))
step = rows[start_idx:]
t0 = int(step[0]["Start_Timestamp"])
with open("$out/last_step.txt", "w") as o:
    for r in step:
        name = r["Kernel_Name"].split("(")[0][:70]
        line = "%8.3f ms  +%7.3f ms  %s  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, name, r.get("Grid_Size", r.get("Grid_Size_X", "")))
        print(line); o.write(line + "\n")
    print("span %.3f ms" % ((max(int(r["End_Timestamp"]) for r in step) - t0) / 1e6))
PY
