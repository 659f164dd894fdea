#!/usr/bin/env python3
"""Development: only bench.py's configs[4] secondary lines (reversed MFAs).  usage: config5_lines.py [n_strings]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from mfa_amd import capi, corpus
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
for line in bench.secondary_config5(torch.device("cuda", 0), capi, corpus, n_strings=n):
    line.pop("parity_sample", None)
    print(json.dumps(line))
