#!/usr/bin/env python3
"""Per-iteration trace of the first strings of a launch (MFA_STATS=1 MFA_STATS_FILE=f run_one.py ...): trace_string.py f [sid]
columns: iteration, scan index i at its start, phase (0 walking, 1 plain periods of a probe, 2 dual period), period q found
by the look at this position, period pp of the probe, plain periods done, D = the wave executed a dual step."""
import sys
import numpy as np
st = np.fromfile(sys.argv[1], dtype=np.uint32)
sid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tr = st[(1 << 20) - 65536 + sid * 16384:][:16384].reshape(-1, 2)
prev = None
for k, (i, w) in enumerate(tr):
    if not (w >> 31): break
    jump = "" if prev is None or i == prev + 1 else "   <- jumped %d" % (int(i) - int(prev) - 1)
    print("%4d i=%-7d phase %d q %d pp %-2d periods %d %s%s" % (k, i, w & 15, (w >> 4) & 15, (w >> 8) & 255, (w >> 16) & 255, "D" if (w >> 24) & 1 else " ", jump))
    prev = int(i)
