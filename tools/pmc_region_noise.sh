#!/bin/bash
# usage: tools/pmc_region_noise.sh   -- SQ counters of region_scan_kernel over 32768 noise strings of 64 KiB (tools/region_64k.py kind "noise"), per row of 1 KiB
R=$PWD
cd /tmp && export TMPDIR=/tmp
k=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  k=$((k+1))
  ONLY_KIND=noise ONLY_LEN=65536 timeout -k 10 300 rocprofv3 --output-format csv --pmc $set -d $R/gpurun_out/pmc_noise/s$k -o c -- python3 $R/tools/region_64k.py 32768 > $R/gpurun_out/pmc_noise_s$k.log 2>&1 || echo "set $k failed: $set"
done
python3 - <<PY
import csv, glob
last = {}
for f in glob.glob("$R/gpurun_out/pmc_noise/*/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "region_scan_kernel" in r["Kernel_Name"]]
    if not rows: continue
    did = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == did: last[r["Counter_Name"]] = last.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows_total = 32768 * 64
for k in sorted(last): print("%-22s %14.0f  per row %8.1f" % (k, last[k], last[k] / rows_total))
PY
