#!/bin/bash
# per-example counters of the specialised kernels on the headline shard (stats builds are compiled on first use)
for ex in ${EXAMPLES:-1 2 3 4 5 6 7 8 9 10}; do
  MFA_STATS=1 MFA_STATS_FILE=/tmp/steps_$ex.bin timeout -k 10 300 python tools/run_one.py --example $ex --strings 125000 --reps 3 2>&1 | grep -E "mfa_hip stats|GB/s" | sed "s/^/ex$ex: /"
  python tools/step_hist.py /tmp/steps_$ex.bin 125000 $ex | head -1 | sed "s/^/ex$ex: /"
done
