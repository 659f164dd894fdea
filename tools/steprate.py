#!/usr/bin/env python3
"""Only bench.py's per-character step-rate lines (non-periodic text of examples 6 and 9, example 1 without a region table).
usage: steprate.py [n_strings=262144]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mfa_amd import capi, corpus
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
for line in bench.secondary_no_regions(torch.device("cuda", 0), capi, corpus, n_strings=n):
    print(json.dumps(line))
