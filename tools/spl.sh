#!/bin/bash
for spl in 1 2 3 4 6 8; do for q in ${QUEUES:-1 4 10}; do
  echo "== strings/lane $spl queues $q"
  MFA_STRINGS_PER_LANE=$spl GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j[\"value\"], j[\"ms_per_step\"]); print({k:round(v[\"kernel_ms\"],2) for k,v in j[\"per_example\"].items()})" || exit 1
done; done
