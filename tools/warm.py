#!/usr/bin/env python3
"""Compile the specialised kernels of the given fixture automata (default: the ten plain examples) into the cache.
usage: warm.py [name ...]     e.g. warm.py ex2_plain ex8_reverse"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd"))
from concurrent.futures import ThreadPoolExecutor
from mfa_amd import capi, image
names = sys.argv[1:] or ["ex%d_plain" % k for k in range(1, 11)]
def one(n):
    t = time.time()
    with open(os.path.join(ROOT, "tests", "golden", "images", n + ".dump")) as f:
        img = capi.Image(image.blob_from_dump(f.read()))
    try: ok = img.specialize()
    except Exception as e: ok = repr(e)
    return n, ok, time.time() - t
with ThreadPoolExecutor(6) as p:
    for n, ok, dt in p.map(one, names): print("%-14s %s %.0f s" % (n, ok, dt))
