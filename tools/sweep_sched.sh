#!/bin/bash
# development: the headline step under different groupings of the region launches / numbers of walk streams
for rl in ${GROUPS_LIST:-3 4,3,2,1 3,3,3,1 4,3,3 5,3,2 4,4,2 3,3,2,2 6,3,1 5,4,1 4,4,1,1}; do for ws in ${WALK_STREAMS:-3}; do
  r=$(timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --region-launches $rl --walk-streams $ws 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms frac %.3f region %.2f' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['region_scan_kernel']['ms_per_step']))")
  echo "region-launches $rl walk-streams $ws: $r"
done; done
