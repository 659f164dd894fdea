#!/bin/bash
# usage: tools/pmc_region.sh   -- SQ counters of region_scan_kernel over the headline shard (one launch, no walks in the timed steps)
R=$PWD
cd /tmp && export TMPDIR=/tmp
k=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
  k=$((k+1))
  timeout -k 10 300 rocprofv3 --output-format csv --pmc $set -d $R/gpurun_out/pmc_region/s$k -o c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --exp region-only --region-launches one > $R/gpurun_out/pmc_region_s$k.log 2>&1 || echo "set $k failed: $set"
done
python3 - <<PY
import csv, glob, collections
last = {}
for f in glob.glob("$R/gpurun_out/pmc_region/*/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "region_scan_kernel" in r["Kernel_Name"]]
    if not rows: continue
    did = max(int(r["Dispatch_Id"]) for r in rows)          # the timed step's launch
    for r in rows:
        if int(r["Dispatch_Id"]) == did: last[r["Counter_Name"]] = last.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(last): print(k, last[k])
PY
