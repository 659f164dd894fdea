#!/usr/bin/env python3
"""Run ONE example's kernel on a synthetic shard (profiling / experiments).
usage: run_one.py --example 1 --strings 20000 [--mode plain] [--reps 3] [--min-len N --max-len N]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd"))
import numpy as np, torch
from mfa_amd import capi, corpus, image

ap = argparse.ArgumentParser()
ap.add_argument("--example", type=int, default=1)
ap.add_argument("--mode", default="plain")
ap.add_argument("--strings", type=int, default=20000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--min-len", type=int, default=1024)
ap.add_argument("--max-len", type=int, default=65536)
ap.add_argument("--random", default="", help="alphabet: i.i.d. random strings of exactly --max-len bytes instead of pumped ones")
a = ap.parse_args()
dev = torch.device("cuda", 0)
sizes = corpus.pump_sizes(a.strings, 0x5EED0004 + a.example, a.min_len, a.max_len)
ws = (np.arange(a.strings) % 2) == 0
if a.random:
    g = torch.Generator(device=dev); g.manual_seed(7)
    alpha = torch.tensor(list(a.random.encode()), dtype=torch.uint8, device=dev)
    data = alpha[torch.randint(0, len(alpha), (a.strings, a.max_len), generator=g, device=dev)]
    b = torch.cat([data.reshape(-1), torch.zeros(64, dtype=torch.uint8, device=dev)])
    o = torch.arange(0, (a.strings + 1) * a.max_len, a.max_len, dtype=torch.int64, device=dev)
else:
    b, o = corpus.device_batch(a.example, sizes, ws, dev)
with open(os.path.join(ROOT, "tests", "golden", "images", "ex%d_%s.dump" % (a.example, a.mode))) as f:
    img = capi.Image(image.blob_from_dump(f.read()))
res = torch.empty(a.strings, dtype=torch.uint8, device=dev)
for r in range(a.reps):
    img.match_tensors(b, o, res)
    ms = img.last_kernel_ms(0)
    print("ex%d %s: %d strings %.1f MB kernel %.2f ms -> %.2f GB/s (kernel kind %d, accepted %d)" % (
        a.example, a.mode, a.strings, int(o[-1]) / 1e6, ms, int(o[-1]) / ms / 1e6, img.info()["last_kernel"], int(res.sum())))
