#!/bin/bash
# usage: tools/pmc_rev8.sh [strings=25000] [tag=r04_walk_ex8rev]  -- SQ counters of the table walk on ex. 8 -reverse pump-only strings (bench.py's
# configs[4] line), per wave-iteration (the iteration count comes from the MFA_WALK_STATS build on the same strings)
n=${1:-25000}; tag=${2:-r04_walk_ex8rev}
R=$PWD
out=$R/gpurun_out/$tag; mkdir -p $out
python3 tools/rev8_run.py $n 2 1 > $out/stats.txt 2>&1 || { cat $out/stats.txt; exit 1; }
cd /tmp && export TMPDIR=/tmp
k=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" "SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  k=$((k+1))
  timeout -k 10 200 rocprofv3 --output-format csv --pmc $set -d $out/s$k -o c -- python3 $R/tools/rev8_run.py $n 1 > $out/pmc_s$k.log 2>&1 || echo "set $k failed: $set"
done
python3 - <<PY
import csv, glob, collections, re
tot = collections.defaultdict(float)
for f in glob.glob("$out/s*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "walk_kernel" in r["Kernel_Name"]: tot[r["Counter_Name"]] += float(r["Counter_Value"])
txt = open("$out/stats.txt").read()
m = re.search(r"wave-iterations (\d+) \(dual (\d+)\), lane steps (\d+)", txt)
it = float(m.group(1)) if m else 1.0
with open("$out/summary.txt", "w") as o:
    o.write("# tools/pmc_rev8.sh $n: walk_kernel (table engine) on ex. 8 -reverse, pump-only strings; counters summed over ONE launch\n")
    o.write(txt)
    for k in sorted(tot): o.write("%s %.0f\n" % (k, tot[k]))
    d = tot
    if m and d.get("SQ_WAVE_CYCLES"):
        o.write("# per wave-iteration (%d iterations): VALU %.0f SALU %.0f branch %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f SMEM %.0f | wave-cycles %.0f, waiting %.0f %%, issuing %.0f %%\n" % (
            it, d["SQ_INSTS_VALU"]/it, d["SQ_INSTS_SALU"]/it, d["SQ_INSTS_BRANCH"]/it, d["SQ_INSTS_LDS"]/it, d["SQ_INSTS_VMEM_RD"]/it, d["SQ_INSTS_VMEM_WR"]/it, d["SQ_INSTS_SMEM"]/it,
            4*d["SQ_WAVE_CYCLES"]/it, 100*d["SQ_WAIT_ANY"]/d["SQ_WAVE_CYCLES"], 100*d["SQ_ACTIVE_INST_ANY"]/d["SQ_WAVE_CYCLES"]))
print(open("$out/summary.txt").read())
PY
