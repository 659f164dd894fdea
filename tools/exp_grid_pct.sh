#!/bin/bash
mkdir -p gpurun_out/exp
for r in 1 2; do for pct in 100 85 75 60 50; do
  MFA_WALK_GRID_PCT=$pct timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('pct=%-4s value %.0f GB/s  ms/step %.3f  frac %.3f  region %.3f ms' % ('$pct', d['value'], d['ms_per_step'], r['frac'], r['region_scan_kernel']['ms_per_step']), flush=True)"
done; done
