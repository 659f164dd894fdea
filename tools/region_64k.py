#!/usr/bin/env python3
"""Development: the region pass alone over strings of one length and one kind (BASELINE configs[2] shapes).
usage: region_64k.py [n_strings] -- prints one line per (kind, length)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from mfa_amd import capi
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
g = torch.Generator(device=dev); g.manual_seed(7)

def run(kind, length):
    flat = torch.full((n * length + 64,), ord("a"), dtype=torch.uint8, device=dev)
    flat[-64:] = 0
    data = flat[:n * length].view(n, length)
    if kind == "last_b":
        data[:, -1] = ord("b")
    elif kind == "one_b":
        data[torch.arange(n, device=dev), torch.randint(0, length, (n,), generator=g, device=dev)] = ord("b")
    elif kind == "noise":
        for lo in range(0, n, 4096):
            hi = min(n, lo + 4096)
            data[lo:hi].masked_fill_(torch.rand((hi - lo, length), generator=g, device=dev) < 0.01, ord("b"))
    off = torch.arange(0, (n + 1) * length, length, dtype=torch.int64, device=dev)
    tab = torch.empty((n, 16), dtype=torch.int64, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record(); capi.region_scan(flat, off, tab); b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev][1:]))
    print("%-7s length %6d x %d: %.3f ms = %.0f GB/s" % (kind, length, n, ms, n * length / ms / 1e6), flush=True)
    del flat, data, off, tab

kinds = [os.environ["ONLY_KIND"]] if os.environ.get("ONLY_KIND") else ["all_a", "last_b", "one_b", "noise"]
lengths = [int(os.environ["ONLY_LEN"])] if os.environ.get("ONLY_LEN") else [65536, 65536 + 192, 65536 - 1000]
for kind in kinds:
    for length in lengths:
        run(kind, length)
