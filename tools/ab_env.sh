#!/bin/bash
# usage: tools/ab_env.sh "<VAR=val ...>" "<VAR=val ...>" [...] -- the headline step (bench.py, no secondaries) under several environments,
# alternating on one box, three rounds ("-" = no extra variables)
mkdir -p gpurun_out/ab
for r in 1 2 3; do
  for spec in "$@"; do
    if [ "$spec" = "-" ]; then e=""; else e="$spec"; fi
    env $e timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 6 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-44s value %.0f GB/s  ms/step %.3f  frac %.3f  kernels %.3f ms  gap %.3f ms' % ('$spec', d['value'], d['ms_per_step'], r['frac'], 1e3 * r.get('kernel_seconds_per_step', 0), d['ms_per_step'] - 1e3 * r.get('kernel_seconds_per_step', 0)), flush=True)"
  done
done
