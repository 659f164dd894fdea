#!/bin/bash
# instruction-cache counters of the table walk on ex. 8 -reverse pump-only strings
n=${1:-25000}; R=$PWD; out=$R/gpurun_out/r04_icache; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
k=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  k=$((k+1))
  timeout -k 10 200 rocprofv3 --output-format csv --pmc $set -d $out/s$k -o c -- python3 $R/tools/rev8_run.py $n 1 > $out/pmc_s$k.log 2>&1 || echo "set $k failed: $set"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$out/s*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "walk_kernel" in r["Kernel_Name"]: tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, "%.0f" % tot[k])
if tot.get("SQC_ICACHE_REQ"): print("icache miss rate %.2f %%" % (100 * tot["SQC_ICACHE_MISSES"] / tot["SQC_ICACHE_REQ"]))
PY
