#!/bin/bash
# Development: the region pass ALONE at lowered occupancy (unused LDS per workgroup), to see how much of what it loses beside the walk
# kernels is the wave slots they take.  256-thread workgroups = one wave per SIMD each: N workgroups per CU = N waves per SIMD.
out=gpurun_out/exp_region_occupancy.txt
mkdir -p gpurun_out; : > $out
for spec in "0 2" "26000 2" "32000 2" "40000 2" "53000 2" "40000 3" "40000 4" "32000 3" "32000 4" "53000 4" "0 2"; do
  set -- $spec
  echo "MFA_REGION_LDS=$1 MFA_REGION_DEPTH=$2" >> $out
  MFA_REGION_LDS=$1 MFA_REGION_DEPTH=$2 python tools/region_time.py 12 >> $out 2>&1 || exit 1
done
cat $out
