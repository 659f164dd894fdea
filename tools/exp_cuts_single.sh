#!/bin/bash
# development: configs[4] lines (one automaton, 1.9 GB) through mfa_match_mixed under different groupings
mkdir -p gpurun_out/exp
for cuts in "" "0.55" "0.5,0.8" "0.45,0.75,0.92" "0.4,0.7,0.9"; do
  echo "== MFA_MIXED_CUTS='$cuts'"
  MFA_MIXED_CUTS="$cuts" timeout -k 10 300 python tools/config5_lines.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    m = d.get('mixed_call', {})
    print('  %-52s per-image %.3f ms (%.3f)  mixed %.3f ms (%.3f)' % (d['workload'][12:], d['region_ms'] + d['walk_ms'], d['frac_of_hbm_peak_on_touched_bytes'], m.get('span_ms', 0), m.get('frac_of_hbm_peak_on_touched_bytes', 0)))"
done
