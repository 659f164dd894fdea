for ex in 2 5 1; do for L in 1024 4096 16384 65536; do timeout -k 10 120 python tools/run_one.py --example $ex --strings 125000 --min-len $L --max-len $((L+1)) --reps 2 2>&1 | tail -1; done; done
