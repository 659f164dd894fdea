#!/usr/bin/env python3
"""Development: the region pass alone on the headline shard (one launch over the whole mixed batch).  usage: region_time.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from mfa_amd import capi, corpus
dev = torch.device("cuda", 0)
parts_b, parts_o, pos_b = [], [], 0
for ex in [2, 5, 3, 8, 9, 10, 6, 4, 1, 7]:
    sizes = corpus.pump_sizes(125000, 0x5EED0004 + ex, 1024, 65536)
    b, o = corpus.device_batch(ex, sizes, (np.arange(125000) % 2) == 0, dev)
    nb = int(o[-1].item()); parts_b.append(b[:nb]); parts_o.append(o[:-1] + pos_b); pos_b += nb
    del b, o
bytes_all = torch.cat(parts_b + [torch.zeros(64, dtype=torch.uint8, device=dev)])
off_all = torch.cat(parts_o + [torch.tensor([pos_b], dtype=torch.int64, device=dev)])
del parts_b, parts_o
tab = torch.empty((off_all.numel() - 1, 16), dtype=torch.int64, device=dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in ev:
    a.record(); capi.region_scan(bytes_all, off_all, tab); b.record()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in ev]
print("region pass alone (%s): best %.3f ms, median %.3f ms = %.0f GB/s (%.3f of peak); checksum %d" % (
    os.environ.get("MFA_LIB_PATH", "default lib")[-24:], min(ms), float(np.median(ms)), pos_b / np.median(ms) / 1e6, pos_b / np.median(ms) / 1e6 / 8000, int(tab[:, 0].sum().item())))
