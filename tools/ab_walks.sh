#!/bin/bash
# usage: tools/ab_walks.sh <libA|default> <libB|default>  -- walk times of both builds on one box: the ten plain examples, ex. 8 -reverse, non-periodic text
mkdir -p gpurun_out/ab
for lib in "$1" "$2" "$1" "$2"; do
  if [ "$lib" = default ]; then unset MFA_LIB_PATH; else export MFA_LIB_PATH=$PWD/$lib; fi
  echo "== lib=$lib"
  timeout -k 10 300 python tools/walk_bench.py table plain rev8 nonper 2>&1 | grep -E "plain sum|ex8 -reverse|non-periodic|no table"
done
