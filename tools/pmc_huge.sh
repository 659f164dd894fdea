#!/bin/bash
# usage: tools/pmc_huge.sh [strings=128] [len=4000]  -- instruction mix, waits and instruction-cache counters of the ex. 8 -reverse kernel
n=${1:-128}; len=${2:-4000}
R=$PWD
cd /tmp && export TMPDIR=/tmp
k=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_FLAT SQ_INSTS_VMEM_WR" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED"; do
  k=$((k+1))
  timeout -k 10 200 rocprofv3 --output-format csv --pmc $set -d $R/gpurun_out/pmc_huge/s$k -o c -- python3 $R/tools/huge_time.py ex8_reverse 8 $n pump 1 $len $len > $R/gpurun_out/pmc_huge_s$k.log 2>&1 || echo "set $k failed: $set"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$R/gpurun_out/pmc_huge/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("mfa_jit_kernel"): tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
PY
