#!/usr/bin/env python3
"""Development: only bench.py's configs[2] secondary line.  usage: config3_line.py [n_strings]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from mfa_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
out = bench.secondary_config3(torch.device("cuda", 0), capi, n_strings=n)
out.pop("parity_sample", None)
print(json.dumps(out, indent=1))
