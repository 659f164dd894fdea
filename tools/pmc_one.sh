#!/bin/bash
# usage: tools/pmc_one.sh <example> <min-len> <max-len>   -- instruction-mix counters of one example's kernel (two PMC passes)
ex=$1; lo=$2; hi=$3
R=$PWD
cd /tmp && export TMPDIR=/tmp
for set in ${PMC_SETS:+"$PMC_SETS"} "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --output-format csv --pmc $set -d $R/gpurun_out/pmc_one/$tag -o c -- python3 $R/tools/run_one.py --example $ex --mode ${MODE:-plain} --strings ${STRINGS:-125000} --min-len $lo --max-len $hi --reps 1 > $R/gpurun_out/pmc_one_$tag.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$R/gpurun_out/pmc_one/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "${KERNEL:-mfa_jit_kernel}" in r["Kernel_Name"]: tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
PY
