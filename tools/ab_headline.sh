#!/bin/bash
# usage: tools/ab_headline.sh <libA.so|default> <libB.so|default> [rounds=3]  -- the headline step (bench.py, no secondaries) with two builds of the library, alternating on one box
A=$1; B=$2; R=${3:-3}
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  for lib in "$A" "$B"; do
    if [ "$lib" = default ]; then unset MFA_LIB_PATH; else export MFA_LIB_PATH=$PWD/$lib; fi
    timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 6 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('lib=%-28s value %.0f GB/s  ms/step %.3f  frac %.3f  region %.3f ms' % ('$lib', d['value'], d['ms_per_step'], r['frac'], r.get('region_scan_kernel',{}).get('ms_per_step',0)), flush=True)"
  done
done
