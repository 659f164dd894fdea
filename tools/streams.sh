#!/bin/bash
for n in ${STREAMS:-2 3 4 5 6 10}; do
  echo "== streams $n"
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --streams $n 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j[\"value\"], j[\"ms_per_step\"]); print({k:round(v[\"kernel_ms\"],2) for k,v in j[\"per_example\"].items()})" || exit 1
done
