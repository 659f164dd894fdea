#!/bin/bash
# headline bench over stream counts and hardware queue counts: STREAMS="3 4" QUEUES="4 8" tools/streams.sh
for q in ${QUEUES:-4}; do for n in ${STREAMS:-2 3 4 5 6 10}; do
  echo "== queues $q streams $n"
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --streams $n 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j[\"value\"], j[\"ms_per_step\"])" || exit 1
done; done
