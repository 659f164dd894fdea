#!/bin/bash
# End-of-round evidence, run on the MI355X box from the repository root:  tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats over the default bench run          -> gpurun_out/<tag>/stats
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over one headline step -> gpurun_out/<tag>/fetch, write
# tools/profile_collect.py <tag> then turns these into the files kept under profiles/.
tag=${1:-r01x}; R=$PWD; out=$R/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --output-format csv --kernel-trace --stats -d $out/stats -o s -- python3 $R/bench.py > $out/bench_stats_run.log 2>&1 || { tail -5 $out/bench_stats_run.log; exit 1; }
grep "^{\"metric\"" $out/bench_stats_run.log > $out/bench.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --output-format csv --pmc $c -d $out/$c -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $out/bench_$c.log 2>&1 || { tail -5 $out/bench_$c.log; exit 1; }
done
find $out -name "*.csv" | head -20
