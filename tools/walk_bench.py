#!/usr/bin/env python3
"""Development: walk-kernel times, one line per case.  usage: walk_bench.py [mode=table|jit] [cases: plain rev8 nonper stats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib
from mfa_amd import capi, image, corpus
dev = torch.device("cuda", 0)
args = sys.argv[1:]
mode = "table"
if args and args[0] in ("table", "jit"):
    mode = args.pop(0)
cases = args or ["plain", "rev8", "nonper"]
os.environ["MFA_WALK"] = mode
# warm-up: clocks up, library loaded
_x = torch.zeros(1 << 28, dtype=torch.uint8, device=dev)
for _ in range(20):
    _x += 1
torch.cuda.synchronize()
del _x

def timed(img, b, o, table, n, reps=3):
    r = torch.empty(n, dtype=torch.uint8, device=dev)
    ms = []
    for _ in range(reps):
        img.match_tensors_regions(b, o, table, r); ms.append(img.last_kernel_ms(0))
    torch.cuda.synchronize()
    return min(ms[1:]), r

def blob_of(name):
    return image.blob_from_dump(oracle_lib.load_dump(name))

if "plain" in cases or "stats" in cases:
    tot = 0.0
    for ex in [2, 5, 3, 8, 9, 10, 6, 4, 1, 7]:
        if "plain" not in cases and ex not in (1, 2):
            continue
        n = 125000
        sizes = corpus.pump_sizes(n, 0x5EED0004 + ex, 1024, 65536)
        b, o = corpus.device_batch(ex, sizes, (np.arange(n) % 2) == 0, dev)
        tab = capi.region_scan(b, o)
        img = capi.Image(blob_of("ex%d_plain" % ex))
        if "stats" in cases:
            os.environ["MFA_WALK_STATS"] = "1"
            img.match_tensors_regions(b, o, tab); torch.cuda.synchronize()
            del os.environ["MFA_WALK_STATS"]
        t, r = timed(img, b, o, tab, n)
        tot += t
        print("%s ex%-2d plain: walk %.3f ms  (%d accepted)" % (mode, ex, t, int(r.sum())), flush=True)
        del b, o, tab
    print("%s plain sum: %.3f ms" % (mode, tot), flush=True)
if "rev8" in cases:
    for n in (25000, 125000):
        sizes = corpus.pump_sizes(n, 0x5EED0005 + 8, 1024, 65536)
        b, o = corpus.device_batch(8, sizes, np.zeros(n, dtype=bool), dev)
        tab = capi.region_scan(b, o)
        img = capi.Image(blob_of("ex8_reverse"))
        if "stats" in cases and n == 25000:
            os.environ["MFA_WALK_STATS"] = "1"
            img.match_tensors_regions(b, o, tab); torch.cuda.synchronize()
            del os.environ["MFA_WALK_STATS"]
        t, r = timed(img, b, o, tab, n, reps=2)
        nb = int(o[-1].item())
        print("%s ex8 -reverse pump only, %d strings: walk %.3f ms = %.1f GB/s (%d accepted)" % (mode, n, t, nb / t / 1e6, int(r.sum())), flush=True)
        del b, o, tab
if "nonper" in cases:
    rng = np.random.default_rng(0x5EED0007)
    length, n_strings = 4096, 262144
    gens = {6: lambda: b"".join((b"a" * int(k) + b"b") for k in rng.integers(1, 24, size=length // 12))[:length],
            9: lambda: b"b" + bytes(rng.choice(list(b"ab"), size=length - 1).tolist())}
    for ex, gen in gens.items():
        base = [gen() for _ in range(64)]
        d64, o64 = oracle_lib.pack(base)
        reps64 = n_strings // 64
        d_b = torch.cat([torch.from_numpy(d64.copy()).to(dev).repeat(reps64), torch.zeros(64, dtype=torch.uint8, device=dev)])
        o64_t = torch.from_numpy(o64.astype(np.int64)).to(dev)
        d_o = torch.cat([(torch.arange(reps64, device=dev, dtype=torch.int64)[:, None] * int(o64[-1]) + o64_t[None, :-1]).reshape(-1),
                         torch.tensor([reps64 * int(o64[-1])], dtype=torch.int64, device=dev)])
        tab = capi.region_scan(d_b, d_o)
        img = capi.Image(blob_of("ex%d_plain" % ex))
        t, r = timed(img, d_b, d_o, tab, n_strings, reps=2)
        nb = int(d_o[-1].item())
        print("%s non-periodic ex%d: walk %.3f ms = %.1f G steps/s" % (mode, ex, t, nb / t / 1e6), flush=True)
        del d_b, d_o, tab
    n1 = 262144
    b, o = corpus.device_batch(1, np.full(n1, 4096, dtype=np.int64), (np.arange(n1) % 2) == 0, dev)
    img = capi.Image(blob_of("ex1_plain"))
    if "stats" in cases:
        os.environ["MFA_WALK_STATS"] = "1"
        img.match_tensors_regions(b, o, None); torch.cuda.synchronize()
        del os.environ["MFA_WALK_STATS"]
    t, r = timed(img, b, o, None, n1, reps=2)
    nb = int(o[-1].item())
    print("%s ex1 no table: walk %.3f ms = %.1f G steps/s" % (mode, t, nb / t / 1e6), flush=True)
