#!/bin/bash
# usage: VARIANTS="A=1 B=2|C=3" [QUEUES="4 8"] tools/variants.sh warm|bench   -- development: generator-knob variants of the headline bench
IFS='|' read -ra V <<< "${VARIANTS:-MFA_GEN_NONE=0}"
for v in "${V[@]}"; do
  if [ "$1" = warm ]; then env $v python tools/warm.py | grep -v True
  else
    for q in ${QUEUES:-4}; do
      for mode in "--sequential" ""; do
      echo "== $v  queues $q $mode"
      env $v GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline $mode $BENCH_ARGS 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j[\"value\"], j[\"ms_per_step\"]); print({k:round(v[\"kernel_ms\"],2) for k,v in j[\"per_example\"].items()})" || exit 1
      done
    done
  fi
done
