#!/usr/bin/env python3
"""MFA_STATS=1 MFA_STATS_FILE=f run_one.py ...; then: step_hist.py f N example  -> distribution of executed steps per string"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd"))
from mfa_amd import corpus
f, n, ex = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
st = np.fromfile(f, dtype=np.uint32)[:n]
sizes = corpus.pump_sizes(n, 0x5EED0004 + ex, 1024, 65536)
ws = (np.arange(n) % 2) == 0
lens = corpus.layout(ex, sizes, ws)["lens"]
print("steps executed per string: mean %.1f median %d p99 %d max %d" % (st.mean(), np.median(st), np.percentile(st, 99), st.max()))
worst = np.argsort(st)[-8:]
for k in worst:
    s = corpus.host_strings(ex, sizes[k:k+1], ws[k:k+1])[0]
    print(k, "len", lens[k], "steps", st[k], "suffix" if ws[k] else "nosuffix", s[:24], b"...", s[-12:])
big = st > 2000
print("strings with > 2000 executed steps:", int(big.sum()), "of", n)
