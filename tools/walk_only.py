#!/usr/bin/env python3
"""Walk kernel alone with a pre-computed region table, repeated: after the first repetition the bytes the walk touches (a few
hundred per string) sit in the Infinity Cache, so the later repetitions show the walk's time without trips to HBM.
usage: walk_only.py --example 1 [--strings 125000] [--reps 6]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd"))
import numpy as np, torch
from mfa_amd import capi, corpus, image

ap = argparse.ArgumentParser()
ap.add_argument("--example", type=int, default=1)
ap.add_argument("--mode", default="plain")
ap.add_argument("--strings", type=int, default=125000)
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda", 0)
sizes = corpus.pump_sizes(a.strings, 0x5EED0004 + a.example, 1024, 65536)
ws = (np.arange(a.strings) % 2) == 0
b, o = corpus.device_batch(a.example, sizes, ws, dev)
with open(os.path.join(ROOT, "tests", "golden", "images", "ex%d_%s.dump" % (a.example, a.mode))) as f:
    img = capi.Image(image.blob_from_dump(f.read()))
tab = capi.region_scan(b, o)
res = torch.empty(a.strings, dtype=torch.uint8, device=dev)
ms = []
for r in range(a.reps):
    img.match_tensors_regions(b, o, tab, res)
    ms.append(img.last_kernel_ms(0))
print("ex%d %s walk only: %s ms" % (a.example, a.mode, " ".join("%.3f" % m for m in ms)))
