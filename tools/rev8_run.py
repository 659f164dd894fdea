#!/usr/bin/env python3
"""bench.py's secondary line "configs[4]: example 8 -reverse, pump only" alone: ex. 8 `-reverse` (77 nodes) on pump-only strings
(seed 0x5EED0005 + 8, log-uniform 1-64 KiB), region table computed once, then the walk kernel `reps` times.
usage: rev8_run.py [strings=125000] [reps=2] [stats=0|1]   (MFA_WALK=table by default)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib
from mfa_amd import capi, image, corpus
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
stats = len(sys.argv) > 3 and sys.argv[3] == "1"
name = sys.argv[4] if len(sys.argv) > 4 else "ex8_reverse"
ex = int("".join(c for c in name.split("_")[0] if c.isdigit()))
os.environ.setdefault("MFA_WALK", "table")
dev = torch.device("cuda", 0)
sizes = corpus.pump_sizes(n, 0x5EED0005 + ex, 1024, 65536)
b, o = corpus.device_batch(ex, sizes, np.zeros(n, dtype=bool), dev)
tab = capi.region_scan(b, o)
img = capi.Image(image.blob_from_dump(oracle_lib.load_dump(name)))
r = torch.empty(n, dtype=torch.uint8, device=dev)
if stats:
    os.environ["MFA_WALK_STATS"] = "1"
    img.match_tensors_regions(b, o, tab, r); torch.cuda.synchronize()
    del os.environ["MFA_WALK_STATS"]
ms = []
for _ in range(reps):
    img.match_tensors_regions(b, o, tab, r); ms.append(img.last_kernel_ms(0))
torch.cuda.synchronize()
nb = int(o[-1].item())
print("%s pump only, %d strings, %.3f GB: walk ms %s = %.1f GB/s (%d accepted)" % (name, n, nb / 1e9, " ".join("%.3f" % m for m in ms), nb / min(ms) / 1e6, int(r.sum())), flush=True)
