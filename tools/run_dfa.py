#!/usr/bin/env python3
"""BASELINE configs[1] alone: (a|b)*abb Thompson NFA, N random 1 KiB strings (profiling / experiments)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
print(bench.secondary_dfa(torch.device("cuda", 0), int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20))
