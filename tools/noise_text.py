#!/usr/bin/env python3
"""Development: text of many medium runs (i.i.d. bytes, a 0.988 / b 0.012, first byte b) under automata that do not die on it (ex. 9 accepts every string
that starts with b): the walk's step rate where the region table overflows.  usage: noise_text.py [example=9] [strings=16384] [length=65536]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib
from mfa_amd import capi, image
ex = int(sys.argv[1]) if len(sys.argv) > 1 else 9
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
L = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0x5EED0021)
data = torch.where(torch.rand((n, L), generator=g, device=dev) < 0.012, ord("b"), ord("a")).to(torch.uint8)
data[:, 0] = ord("b")
flat = torch.cat([data.reshape(-1), torch.zeros(64, dtype=torch.uint8, device=dev)])
off = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
blob = image.blob_from_dump(oracle_lib.load_dump("ex%d_plain" % ex))
for engine in ("table", "jit"):
    os.environ["MFA_WALK"] = engine
    img = capi.Image(blob)
    res = torch.empty(n, dtype=torch.uint8, device=dev)
    ms = []
    for _ in range(3):
        img.match_tensors(flat, off, res); ms.append((img.last_region_ms(0), img.last_kernel_ms(0)))
    torch.cuda.synchronize()
    r, w = ms[-1]
    sample = [bytes(data[k].cpu().numpy().tobytes()) for k in range(0, n, max(1, n // 6))][:6]
    want = oracle_lib.OracleImage(blob).match(sample)
    got = [int(res[k]) for k in range(0, n, max(1, n // 6))][:6]
    print("ex%d noise text, %d x %d bytes, %s engine: region %.2f ms, walk %.2f ms = %.1f G steps/s, accepted %d, oracle sample %s" % (
        ex, n, L, engine, r, w, n * L / w / 1e6, int(res.sum()), "ok" if list(want) == got else "MISMATCH"), flush=True)
