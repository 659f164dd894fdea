#!/usr/bin/env python3
"""Development: only bench.py's "configs[2], secondary variant" line (log-uniform lengths).  usage: config3lu_line.py [n_strings]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from mfa_amd import capi, corpus
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
line = bench.secondary_config3_loguniform(torch.device("cuda", 0), capi, corpus, n_strings=n)
line.pop("parity_sample", None)
print(json.dumps(line))
