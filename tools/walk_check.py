#!/usr/bin/env python3
"""Development: the table-driven walk (MFA_WALK=table) against goldens and the oracle on every memory automaton, then its time
beside the generated kernels' on the headline shard, example by example, and the mixed batch in one call."""
import glob, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib
from mfa_amd import capi, image, corpus

dev = torch.device("cuda", 0)

def gpu(img, strings):
    data, off = oracle_lib.pack(strings)
    d = torch.zeros(len(data) + 64, dtype=torch.uint8, device=dev); d[:len(data)] = torch.from_numpy(data.copy())
    o = torch.from_numpy(off.astype(np.int64)).to(dev)
    r = img.match_tensors(d, o); torch.cuda.synchronize()
    return r.cpu().numpy()

def parity():
    os.environ["MFA_WALK"] = "table"
    man = json.load(open(os.path.join(oracle_lib.GOLDEN, "manifest.json")))
    bad = tot = 0
    for auto in man["automata"]:
        blob = image.blob_from_dump(oracle_lib.load_dump(auto["name"]))
        if image.blob_info(blob)["kind"] != image.KIND_MFA:
            continue
        img = capi.Image(blob)
        for sset in auto["sets"]:
            strings = oracle_lib.load_set(sset); want = oracle_lib.load_bits(auto["name"], sset)
            got = gpu(img, strings)
            m = int((got != want).sum()); tot += len(strings); bad += m
            if m:
                k = int(np.nonzero(got != want)[0][0])
                print("MISMATCH", auto["name"], sset, m, strings[k][:50], want[k], got[k], flush=True)
        assert img.info()["last_kernel"] == capi.KERNEL_WALK
        name = auto["name"]
        ex = int("".join(c for c in name.split("_")[0] if c.isdigit())) if name.startswith("ex") else None
        if ex in corpus.EXAMPLES:
            sizes = np.array([700, 5000, 20000, 64000, 65536, 33333]); ws = np.array([0, 1, 0, 1, 0, 1], dtype=bool)
            strings = corpus.host_strings(ex, sizes, ws)
            rng = np.random.default_rng(ex)
            extra = []
            for s in strings[:4]:
                b = bytearray(s); b[int(rng.integers(len(b)))] = ord("b"); extra.append(bytes(b))
                extra.append(s[:len(s) // 2] + s[len(s) // 3:])
            strings += extra
            want = oracle_lib.OracleImage(blob).match(strings)
            got = gpu(img, strings)
            m = int((got != want).sum()); tot += len(strings); bad += m
            if m:
                print("MISMATCH long", name, m, flush=True)
        img.close()
    print("parity: %d strings, %d mismatches" % (tot, bad), flush=True)
    return bad

def timing(n_per=125000):
    layout = [2, 5, 3, 8, 9, 10, 6, 4, 1, 7]
    parts_b, parts_o, seg, pos_b = [], [], [0], 0
    imgs = {}
    for ex in layout:
        sizes = corpus.pump_sizes(n_per, 0x5EED0004 + ex, 1024, 65536)
        ws = (np.arange(n_per) % 2) == 0
        b, o = corpus.device_batch(ex, sizes, ws, dev)
        nb = int(o[-1].item())
        blob = image.blob_from_dump(oracle_lib.load_dump("ex%d_plain" % ex))
        res = {}
        for mode in (() if os.environ.get("SKIP_PER_EX") else ("jit", "table")):
            os.environ["MFA_WALK"] = mode
            img = capi.Image(blob)
            r = torch.empty(n_per, dtype=torch.uint8, device=dev)
            ms = []
            for _ in range(3):
                img.match_tensors(b, o, r); ms.append((img.last_kernel_ms(0), img.last_region_ms(0)))
            torch.cuda.synchronize()
            res[mode] = (ms[-1], r.clone())
            imgs.setdefault(ex, img) if mode == "table" else None
        if res:
          same = bool(torch.equal(res["jit"][1], res["table"][1]))
          print("ex%-2d %6.0f MB  jit walk %.3f ms  table walk %.3f ms  region %.3f ms  same=%s accepted=%d" % (
            ex, nb / 1e6, res["jit"][0][0], res["table"][0][0], res["table"][0][1], same, int(res["table"][1].sum())), flush=True)
        parts_b.append(b[:nb]); parts_o.append(o[:-1] + pos_b); pos_b += nb; seg.append(seg[-1] + n_per)
        del b, o
    bytes_all = torch.cat(parts_b + [torch.zeros(64, dtype=torch.uint8, device=dev)])
    off_all = torch.cat(parts_o + [torch.tensor([pos_b], dtype=torch.int64, device=dev)])
    del parts_b, parts_o
    # reference answers: every example by itself, generated kernels
    os.environ["MFA_WALK"] = "jit"
    ref = torch.zeros(seg[-1], dtype=torch.uint8, device=dev)
    for k, ex in enumerate(layout):
        img = capi.Image(image.blob_from_dump(oracle_lib.load_dump("ex%d_plain" % ex)))
        a, b2 = seg[k], seg[k + 1]
        img.match_tensors(bytes_all, off_all[a:b2 + 1], ref[a:b2]); torch.cuda.synchronize()
    for engine in os.environ.get("ENGINES", "jit,table").split(","):
        os.environ["MFA_WALK"] = engine
        ims = [capi.Image(image.blob_from_dump(oracle_lib.load_dump("ex%d_plain" % ex))) for ex in layout]
        for groups in os.environ.get("CUTS", "default").split("|"):
            if groups == "default":
                os.environ.pop("MFA_MIXED_CUTS", None)
            else:
                os.environ["MFA_MIXED_CUTS"] = groups
            mx = capi.Mixed(ims)
            res = torch.zeros(seg[-1], dtype=torch.uint8, device=dev)
            t = []
            for _ in range(8):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                mx.match_tensors(bytes_all, off_all, seg, res); torch.cuda.synchronize()
                t.append((time.perf_counter() - t0) * 1e3)
            r_ms, sp_ms = mx.last_ms(0)
            best = min(t[2:])
            print("mixed %-5s cuts=%s: wall %.3f ms (best of 6)  regions done at %.3f  span %.3f  -> %.0f GB/s  frac %.3f  same=%s" % (
                engine, groups, best, r_ms, sp_ms, pos_b / (best * 1e-3) / 1e9, pos_b / (best * 1e-3) / 1e9 / 8000, bool(torch.equal(res, ref))), flush=True)
            mx.close()

if __name__ == "__main__":
    what = sys.argv[1:] or ["parity", "timing"]
    rc = 0
    if "parity" in what:
        rc = parity()
    if "timing" in what:
        timing()
    sys.exit(1 if rc else 0)
