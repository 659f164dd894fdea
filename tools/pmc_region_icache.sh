#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $R/gpurun_out/pmc_ic -o c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --exp region-only --region-launches one > $R/gpurun_out/pmc_ic.log 2>&1
python3 - <<PY
import csv, glob
last={}
for f in glob.glob("$R/gpurun_out/pmc_ic/*counter_collection.csv")+glob.glob("$R/gpurun_out/pmc_ic/*/*counter_collection.csv"):
    rows=[r for r in csv.DictReader(open(f)) if "region_scan_kernel" in r["Kernel_Name"]]
    did=max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"])==did: last[r["Counter_Name"]]=last.get(r["Counter_Name"],0)+float(r["Counter_Value"])
for k in sorted(last): print(k,last[k])
PY
