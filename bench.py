#!/usr/bin/env python3
"""Benchmark of the match path: input GB/s on the 10-example attack corpus (BASELINE.json metric).

One step = one pass of the hot path over this rank's shard of the mixed corpus (BASELINE configs[3]): ONE batch
(one byte buffer, one offset array) that holds the strings of all ten README examples, example after example.
Per step and per example: one launch of the region pre-pass (region_scan_kernel, mfa_region_scan) on the region
stream and one launch of the example's walk kernel (mfa_match_batch_regions) on a walk stream that waits for it;
then the result bitmap of the shard is gathered to rank 0 (RCCL when N > 1).
Per-GPU work is fixed (weak scaling): rank r holds `--strings-per-example` strings of every example,
`prefix + pumped_string(n, pump) [+ suffix]` with n log-uniform in [--min-len, --max-len]
(generator: reference matchers/example_runner.cpp:15-29), generated on the device before the timed
region, so the timed region starts with all inputs resident in HBM.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (fresh
processes, before anything touches a GPU); under torchrun (RANK / WORLD_SIZE set) it is one of the ranks:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
PARITY_N = 48                  # strings per example whose GPU answers are re-checked on the CPU after the timed region
GOLDEN = os.path.join(ROOT, "tests", "golden")
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r02f_traffic.json")


def spawn_ranks(args):
    """`--gpus N` without a launcher: start N fresh processes, one per GPU, and wait for them.  Nothing in this process
    has touched a GPU (counting devices does not initialise the runtime), and no process is replaced by exec."""
    import torch
    n_dev = torch.cuda.device_count()
    env = dict(os.environ)
    if n_dev < args.gpus and env.get("MFA_BENCH_REHEARSE") != "1":
        sys.stderr.write("bench.py: --gpus %d but %d device(s) visible; MFA_BENCH_REHEARSE=1 runs the ranks on one GPU over gloo\n"
                         % (args.gpus, n_dev))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env.update({"WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "MFA_BENCH_SPAWNED": "1"})
    procs = []
    for r in range(args.gpus):
        e = dict(env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def load_blob(name):
    from mfa_amd import image
    with open(os.path.join(GOLDEN, "images", name + ".dump")) as f:
        return image.blob_from_dump(f.read())


# ---- CPU baseline (rank 0, N = 1 only; bounded sample) ------------------------------------------------------------
def _oracle_time(args):
    cli, blob_path, text = args
    p = subprocess.run([cli, "time", blob_path], input=text, capture_output=True)
    if p.returncode != 0:
        return None
    f = p.stdout.split()
    return int(f[0]), int(f[1]), float(f[2]), int(f[3])


def cpu_baseline(corpus, shards, gpu_results, budget_strings=8, cap=32768):
    """Time the reference itself (oracle/_ref/ref_harness: its own sources, built as it builds them, no
    -O flag, canonical allocation-order mode) on a bounded sample of the same workload; falls back to
    our CPU restatement when the reference build is not present.  Beside it: the restatement on ALL host
    cores (one process per core over the parity sample), core count stated."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    cli = os.path.join(ROOT, "oracle", "oracle_cli")
    use_ref = os.path.exists(ref)
    if not use_ref and not os.path.exists(cli):
        return None
    tot_bytes, tot_sec, n_str, acc_cpu, acc_gpu = 0, 0.0, 0, 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        for ex, sh in shards.items():
            # the shortest-but-representative sample: first strings of the shard, capped in length so the
            # whole baseline stays within ~10-30 s (the reference is ~quadratic in string length)
            idx = [k for k, s in enumerate(sh["sample"]) if len(s) <= cap][:budget_strings]
            sample = [sh["sample"][k] for k in idx]
            if not sample:
                continue
            acc_gpu += int(sum(int(gpu_results[ex][k]) for k in idx))
            text = b"".join(s + b"\n" for s in sample)
            if use_ref:
                cmd = [ref, "time", "plain", corpus.EXAMPLES[ex][0]]
            else:
                blob_path = os.path.join(tmp, "ex%d.blob" % ex)
                with open(blob_path, "wb") as f:
                    f.write(sh["blob"])
                cmd = [cli, "time", blob_path]
            p = subprocess.run(cmd, input=text, capture_output=True, cwd=tmp)
            if p.returncode != 0:
                return None
            f = p.stdout.split()
            n_str += int(f[0]); tot_bytes += int(f[1]); tot_sec += float(f[2]); acc_cpu += int(f[3])
        # wider parity check with the CPU restatement (fast): the first PARITY_N strings of every example
        checked, mism, port_bytes, port_sec = 0, 0, 0, 0.0
        all_cores = None
        if os.path.exists(cli):
            jobs = []
            for ex, sh in shards.items():
                sample = [s for s in sh["sample"] if len(s) <= 32768]
                idx = [k for k, s in enumerate(sh["sample"]) if len(s) <= 32768]
                blob_path = os.path.join(tmp, "p%d.blob" % ex)
                with open(blob_path, "wb") as f:
                    f.write(sh["blob"])
                t0 = time.perf_counter()
                p = subprocess.run([cli, "match", blob_path], input=b"".join(s + b"\n" for s in sample), capture_output=True, cwd=tmp)
                if p.returncode != 0:
                    continue
                port_sec += time.perf_counter() - t0
                port_bytes += sum(len(s) for s in sample)
                want = [int(x) for x in p.stdout.split()]
                got = [int(gpu_results[ex][k]) for k in idx]
                checked += len(want)
                mism += sum(1 for a, b in zip(want, got) if a != b) + abs(len(want) - len(got))
                jobs.append((blob_path, sample))
            # the restatement on every host core: the parity sample dealt round-robin to one process per core
            # this process's CPU share (the GPU box gives one GPU's share of the host, not all of its cores)
            cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16))
            from concurrent.futures import ThreadPoolExecutor
            work = []
            for blob_path, sample in jobs:
                for c in range(cores):
                    part = sample[c::cores]
                    if part:
                        work.append((cli, blob_path, b"".join(s + b"\n" for s in part)))
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as pool:
                done = [d for d in pool.map(_oracle_time, work) if d]
            wall = time.perf_counter() - t0
            if done and wall > 0:
                all_cores = {"value": sum(d[1] for d in done) / wall / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
                             "sample": "%d strings, %d bytes, %.2f s wall, one oracle_cli process per core (process start included)" % (
                                 sum(d[0] for d in done), sum(d[1] for d in done), wall)}
    if tot_sec <= 0:
        return None
    return {"value": tot_bytes / tot_sec / 1e9, "unit": "GB/s", "cores": 1, "kind": "reference" if use_ref else "port",
            "sample": "%d strings (first <=%d of each example's shard with length <= %d KiB), %d bytes, %.1f s, 1 thread" % (
                n_str, budget_strings, cap // 1024, tot_bytes, tot_sec),
            "provenance": ("oracle/_ref/ref_harness: /root/reference's own .cpp files compiled where they lie by oracle/Makefile, no -O flag "
                           "(as its CMakeLists builds them), bump-arena allocation-order mode" if use_ref else
                           "oracle/oracle_cli: CPU restatement oracle/mfa_oracle.c, -O2"),
            # the same strings were matched on the GPU in the timed region: the accept counts must agree
            "accepted": acc_cpu, "accepted_gpu": acc_gpu, "parity": acc_cpu == acc_gpu and mism == 0,
            "parity_restatement": {"strings": checked, "mismatches": mism},
            # the CPU restatement (oracle/mfa_oracle.c, -O2, one thread) on the parity sample, process start included
            "restatement": {"value": port_bytes / port_sec / 1e9 if port_sec > 0 else None, "unit": "GB/s", "cores": 1, "kind": "port",
                            "sample": "%d strings, %d bytes, %.2f s" % (checked, port_bytes, port_sec)},
            "restatement_all_cores": all_cores}


# ---- secondary lines (other BASELINE configs; rank 0, N = 1 only) --------------------------------------------------
def _timed(img, flat, off, res, device, reps=3):
    import numpy as np
    import torch
    ms, rms = [], []
    for _ in range(reps + 1):
        img.match_tensors(flat, off, res)
        ms.append(img.last_kernel_ms(device.index or 0))
        rms.append(img.last_region_ms(device.index or 0))
    torch.cuda.synchronize()
    return float(np.mean(ms[1:])), float(np.mean(rms[1:]))


def _oracle_sample(blob, strings, got):
    """parity field of a secondary line: the CPU restatement on a sample of the line's own strings"""
    import oracle_lib
    want = oracle_lib.OracleImage(blob).match(strings)
    bad = sum(1 for a, b in zip(want, got) if int(a) != int(b))
    return {"strings": len(strings), "bytes": sum(len(s) for s in strings), "mismatches": bad}


def secondary_dfa(device, capi, n_strings=1 << 20, length=1024):
    """BASELINE.json configs[1]: Thompson automaton of (a|b)*abb through the API, 1M random 1 KiB strings."""
    import torch
    img = capi.Image(load_blob("nfa_abb_thompson"))
    g = torch.Generator(device=device); g.manual_seed(0x5EED0002)
    data = torch.randint(0, 2, (n_strings, length), generator=g, device=device, dtype=torch.uint8) + ord("a")
    data[0::2, -3:] = torch.tensor(list(b"abb"), dtype=torch.uint8, device=device)
    flat = torch.cat([data.reshape(-1), torch.zeros(64, dtype=torch.uint8, device=device)])
    off = torch.arange(0, (n_strings + 1) * length, length, dtype=torch.int64, device=device)
    res = torch.empty(n_strings, dtype=torch.uint8, device=device)
    t, _ = _timed(img, flat, off, res, device)
    want = (data[:, -3] == ord("a")) & (data[:, -2] == ord("b")) & (data[:, -1] == ord("b"))
    ok = bool(torch.equal(res.bool(), want))
    gbs = n_strings * length / (t * 1e-3) / 1e9
    return {"workload": "configs[1]: (a|b)*abb Thompson NFA (tabulated), %d random %d-byte strings" % (n_strings, length),
            "kernel": "dfa_tiled_kernel", "kernel_ms": t, "GB/s": gbs, "touched_bytes": n_strings * length,
            "frac_of_hbm_peak_on_touched_bytes": gbs / HBM_PEAK_GBS, "note": "no early exit: every byte of every string is walked",
            "results_exact": ok}


def secondary_config3(device, capi, n_strings=1 << 17, length=65536):
    """BASELINE.json configs[2] (roofline variant of SURVEY section 8d): example 1, strings of exactly 64 KiB, mix by
    j mod 4: a^(L-1) b, a^L, a^L with one byte at a seeded position set to b, i.i.d. {a: 0.99, b: 0.01}."""
    import torch
    blob = load_blob("ex1_plain")
    img = capi.Image(blob)
    g = torch.Generator(device=device); g.manual_seed(0x5EED0003)
    data = torch.full((n_strings, length), ord("a"), dtype=torch.uint8, device=device)
    data[0::4, -1] = ord("b")
    rows = torch.arange(2, n_strings, 4, device=device)
    data[rows, torch.randint(0, length, (rows.numel(),), generator=g, device=device)] = ord("b")
    for lo in range(3, n_strings, 4 * 4096):                      # the random rows, in chunks
        r = torch.arange(lo, min(n_strings, lo + 4 * 4096), 4, device=device)
        mask = torch.rand((r.numel(), length), generator=g, device=device) < 0.01
        data[r] = torch.where(mask, torch.tensor(ord("b"), dtype=torch.uint8, device=device), data[r])
        del mask
    short = [bytes(data[k, :3000].cpu().numpy().tobytes()) for k in range(8)]
    flat = torch.cat([data.reshape(-1), torch.zeros(64, dtype=torch.uint8, device=device)])
    del data
    off = torch.arange(0, (n_strings + 1) * length, length, dtype=torch.int64, device=device)
    res = torch.empty(n_strings, dtype=torch.uint8, device=device)
    t, tr = _timed(img, flat, off, res, device)
    nbytes = n_strings * length
    # the region pass stops reading a string once its table is full (64 candidates: text made of hundreds of medium runs, here the
    # a/b noise strings): those strings' bytes are only partly touched
    tab = capi.region_scan(flat, off)
    cut = int(((tab[:, 0] & capi.REGION_OVERFLOW) != 0).sum().item())
    del tab
    # a^L is accepted (SURVEY section 8c anchors: aa, aaa, aaaa, aaaaaaaa -> 1), every string containing a b is not
    ok = bool(res[1::4].all().item()) and not bool(res[0::4].any().item()) and not bool(res[2::4].any().item()) and not bool(res[3::4].any().item())
    import oracle_lib
    data_s, off_s = oracle_lib.pack(short)
    import numpy as np
    d_b = torch.zeros(len(data_s) + 64, dtype=torch.uint8, device=device); d_b[:len(data_s)] = torch.from_numpy(data_s.copy())
    got = img.match_tensors(d_b, torch.from_numpy(off_s.astype(np.int64)).to(device)).cpu().numpy()
    return {"workload": "configs[2]: ({a*}:1&1)*, %d strings of exactly %d bytes, 4-way attack mix" % (n_strings, length),
            "kernel": "region_scan_kernel + mfa_jit_kernel", "region_ms": tr, "walk_ms": t, "GB/s_on_sum_of_lengths": nbytes / ((t + tr) * 1e-3) / 1e9,
            # the walk stops at the first empty state set (mfa.cpp:224-225), but the region pass has read every byte by then
            "touched_bytes": {"at_least": nbytes - cut * length, "at_most": nbytes},
            "touched_by": "region_scan_kernel reads every byte of every string except %d strings (%.1f %%) whose region table filled up, which it "
                          "stops reading there; the walk reads only the bytes of the steps it executes" % (cut, 100.0 * cut / n_strings),
            "frac_of_hbm_peak_on_touched_bytes": {"at_least": (nbytes - cut * length) / ((t + tr) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                  "at_most": nbytes / ((t + tr) * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "region_pass_GB/s": nbytes / (tr * 1e-3) / 1e9 if tr > 0 else None,
            "results_as_expected": ok, "parity_oracle_sample": _oracle_sample(blob, short, got)}


def secondary_64k_all_examples(device, capi, corpus, n_strings=16384):
    """north_star: "at 64 KiB strings" -- all ten examples, every string pumped to 64 KiB (pump size 65536), alternating
    with / without suffix; per example region pass + walk, back to back."""
    import numpy as np
    import torch
    out = {"workload": "north_star size: 10 examples, %d strings each, pump size exactly 65536, alternating with/without suffix" % n_strings,
           "kernel": "region_scan_kernel + mfa_jit_kernel, per example, back to back on one stream", "per_example": {}}
    tot_b, tot_ms = 0, 0.0
    for ex in sorted(corpus.EXAMPLES):
        img = capi.Image(load_blob("ex%d_plain" % ex))
        sizes = np.full(n_strings, 65536, dtype=np.int64)
        flat, off = corpus.device_batch(ex, sizes, (np.arange(n_strings) % 2) == 0, device)
        res = torch.empty(n_strings, dtype=torch.uint8, device=device)
        t, tr = _timed(img, flat, off, res, device, reps=2)
        nb = int(off[-1].item())
        out["per_example"][str(ex)] = {"bytes": nb, "region_ms": tr, "walk_ms": t, "GB/s": nb / ((t + tr) * 1e-3) / 1e9, "accepted": int(res.sum().item())}
        tot_b += nb; tot_ms += t + tr
        del flat, off, res
    out["GB/s"] = tot_b / (tot_ms * 1e-3) / 1e9
    out["touched_bytes"] = tot_b
    out["frac_of_hbm_peak_on_touched_bytes"] = out["GB/s"] / HBM_PEAK_GBS
    return out


def secondary_no_regions(device, capi, corpus, n_strings=262144, length=4096):
    """The per-character step rate as a tracked number: in-language text WITHOUT long periodic regions (nothing for the
    acceleration to skip): example 1's language is every a^n, so a non-periodic in-language text does not exist for it; examples
    6 and 9 accept texts with irregular block lengths.  Also example 1 on its attack strings with MFA_ACCEL=0 semantics (no
    region table: every step executed)."""
    import numpy as np
    import torch
    import oracle_lib
    rng = np.random.default_rng(0x5EED0007)
    lines = []
    # ex 6 ({a*}:1b|&1)*: blocks a^k b with the same k repeated are in the language; irregular k keeps regions short
    # ex 9 (({aa*b}:1(&1)*)|b(b|a*)*)*: b (b|a*)* accepts every string over {a, b} that starts with b
    gens = {6: lambda: b"".join((b"a" * int(k) + b"b") for k in rng.integers(1, 24, size=length // 12))[:length],
            9: lambda: b"b" + bytes(rng.choice(list(b"ab"), size=length - 1).tolist())}
    for ex, gen in gens.items():
        blob = load_blob("ex%d_plain" % ex)
        img = capi.Image(blob)
        base = [gen() for _ in range(64)]
        # the batch is the 64 texts repeated (string k = text k % 64), built on the device: enough strings to fill every CU
        d64, o64 = oracle_lib.pack(base)
        reps64 = n_strings // 64
        d_b = torch.cat([torch.from_numpy(d64.copy()).to(device).repeat(reps64), torch.zeros(64, dtype=torch.uint8, device=device)])
        o64_t = torch.from_numpy(o64.astype(np.int64)).to(device)
        d_o = torch.cat([(torch.arange(reps64, device=device, dtype=torch.int64)[:, None] * int(o64[-1]) + o64_t[None, :-1]).reshape(-1),
                         torch.tensor([reps64 * int(o64[-1])], dtype=torch.int64, device=device)])
        off = d_o
        res = torch.empty(n_strings, dtype=torch.uint8, device=device)
        t, tr = _timed(img, d_b, d_o, res, device, reps=2)
        nb = int(off[-1].item())
        short = [s[:1500] for s in base[:8]]
        ds, os_ = oracle_lib.pack(short)
        d_s = torch.zeros(len(ds) + 64, dtype=torch.uint8, device=device); d_s[:len(ds)] = torch.from_numpy(ds.copy())
        got = img.match_tensors(d_s, torch.from_numpy(os_.astype(np.int64)).to(device)).cpu().numpy()
        lines.append({"workload": "non-periodic text, example %d, %d strings of %d bytes (64 distinct)" % (ex, n_strings, length),
                      "kernel": "region_scan_kernel + mfa_jit_kernel", "region_ms": tr, "walk_ms": t, "GB/s": nb / ((t + tr) * 1e-3) / 1e9,
                      "char_steps_per_s": nb / ((t + tr) * 1e-3), "accepted": int(res.sum().item()),
                      "parity_oracle_sample": _oracle_sample(blob, short, got)})
        del d_b, d_o, res
    # example 1, attack strings, no table: every step is executed
    img = capi.Image(load_blob("ex1_plain"))
    n1 = 262144
    sizes = np.full(n1, 4096, dtype=np.int64)
    flat, off = corpus.device_batch(1, sizes, (np.arange(n1) % 2) == 0, device)
    res = torch.empty(n1, dtype=torch.uint8, device=device)
    ms = []
    for _ in range(3):
        img.match_tensors_regions(flat, off, None, res)
        ms.append(img.last_kernel_ms(device.index or 0))
    nb = int(off[-1].item())
    t = float(np.mean(ms[1:]))
    lines.append({"workload": "example 1, %d attack strings of 4 KiB, NO region table (every step executed)" % n1, "kernel": "mfa_jit_kernel",
                  "walk_ms": t, "GB/s": nb / (t * 1e-3) / 1e9, "char_steps_per_s": nb / (t * 1e-3), "accepted": int(res.sum().item())})
    return lines


def secondary_config5(device, capi, corpus, n_strings=125000):
    """BASELINE.json configs[4]: reversed MFAs (`-reverse`, is_reversed = 1) of the nondeterministic examples 3, 6, 8 on
    pump-only strings (full walk) and pump+suffix strings (early exit), reported separately."""
    import numpy as np
    import torch
    out = []
    for ex in (3, 6, 8):
        blob = load_blob("ex%d_reverse" % ex)
        img = capi.Image(blob)
        n = n_strings if ex != 8 else n_strings // 5
        for tag, suffix in (("pump only", False), ("pump + suffix", True)):
            sizes = corpus.pump_sizes(n, 0x5EED0005 + ex, 1024, 65536)
            ws = np.full(n, suffix)
            flat, off = corpus.device_batch(ex, sizes, ws, device)
            res = torch.empty(n, dtype=torch.uint8, device=device)
            t, tr = _timed(img, flat, off, res, device, reps=2)
            nbytes = int(off[-1].item())
            short = [k for k in range(n) if sizes[k] <= 2500][:24]
            strings = corpus.host_strings(ex, sizes[short], ws[short])
            out.append({"workload": "configs[4]: example %d -reverse, %d strings, %s" % (ex, n, tag),
                        "kernel": {capi.KERNEL_GENERIC: "mfa_walk_kernel", capi.KERNEL_SPECIALISED: "region_scan_kernel + mfa_jit_kernel"}[img.info()["last_kernel"]],
                        "region_ms": tr, "walk_ms": t, "GB/s": nbytes / ((t + tr) * 1e-3) / 1e9,
                        "touched_bytes": nbytes, "touched_by": "region pass reads every byte; the walk may exit early",
                        "frac_of_hbm_peak_on_touched_bytes": nbytes / ((t + tr) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "accepted": int(res.sum().item()),
                        "parity_oracle_sample": _oracle_sample(blob, strings, res[short].cpu().numpy())})
            del flat, off, res
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--strings-per-example", type=int, default=125000,
                    help="strings of each example per GPU (125000 x 10 examples x 8 GPUs = the 10M-string batch)")
    ap.add_argument("--min-len", type=int, default=1024)
    ap.add_argument("--max-len", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--walk-streams", type=int, default=3, help="HIP streams the ten walk launches of a step are spread over")
    ap.add_argument("--region-streams", type=int, default=1, help="HIP streams the region launches alternate between (the tail of one launch overlaps the start of the next)")
    ap.add_argument("--region-priority", type=int, default=0, help="stream priority of the region streams (-1 = high)")
    ap.add_argument("--region-launches", default="3",
                    help="region pre-pass launches per step: 'per-example' (10), 'one', explicit group sizes 'a,b,c', or a number g: the examples, laid out in the batch "
                         "costliest walk first, are scanned in g launches of consecutive examples; the walks of a group start when "
                         "their group is scanned and run beside the next group's scan")
    ap.add_argument("--walk-waves", type=int, default=0, help="development: cap the walk kernels at this many waves per CU (MFA_WALK_WAVES_PER_CU)")
    ap.add_argument("--exp", default="", help="development: 'region-only' skips the walk launches")
    ap.add_argument("--order", default="", help="comma-separated order of the examples' segments in the batch (default: costliest walk first)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch
    from mfa_amd import capi, corpus, sharding
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if args.walk_waves > 0:
        os.environ["MFA_WALK_WAVES_PER_CU"] = str(args.walk_waves)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: using WORLD_SIZE\n" % (args.gpus, world))
    # MFA_BENCH_REHEARSE=1: N ranks on ONE GPU with the gloo backend (bitmaps gathered through host memory) -- only to
    # exercise the N > 1 code path on a single-GPU box; real runs use one GPU per rank and RCCL
    rehearse = os.environ.get("MFA_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    comm_dev = torch.device("cpu") if rehearse else device

    # ---- this rank's shard: ONE mixed batch, resident in HBM ------------------------------------------
    shards = {}
    n_per = args.strings_per_example
    parts_b, parts_o, pos_s, pos_b = [], [], 0, 0
    # segment order in the batch: costliest walk first (measured once on MI355X; any order is correct)
    layout = [int(x) for x in args.order.split(",")] if args.order else [2, 5, 3, 8, 9, 10, 6, 4, 1, 7]
    assert sorted(layout) == sorted(corpus.EXAMPLES)
    for ex in layout:
        seed = 0x5EED0004 + 1000 * rank + ex
        sizes = corpus.pump_sizes(n_per, seed, args.min_len, args.max_len)
        with_suffix = (np.arange(n_per) % 2) == 0
        d_bytes, d_off = corpus.device_batch(ex, sizes, with_suffix, device)
        nbytes = int(d_off[-1].item())
        blob = load_blob("ex%d_plain" % ex)
        img = capi.Image(blob)
        img.prepare(local)
        parts_b.append(d_bytes[:nbytes])
        parts_o.append(d_off[:-1] + pos_b)
        shards[ex] = {"img": img, "n": n_per, "nbytes": nbytes, "blob": blob, "first": pos_s,
                      "sample": corpus.host_strings(ex, sizes[:PARITY_N], with_suffix[:PARITY_N]) if rank == 0 else []}
        pos_s += n_per
        pos_b += nbytes
        del d_bytes, d_off
    total_strings, total_bytes = pos_s, pos_b
    bytes_all = torch.cat(parts_b + [torch.zeros(64, dtype=torch.uint8, device=device)])
    off_all = torch.cat(parts_o + [torch.tensor([total_bytes], dtype=torch.int64, device=device)])
    del parts_b, parts_o
    torch.cuda.empty_cache()
    table = torch.empty((total_strings, capi.REGION_WORDS), dtype=torch.int64, device=device)
    results = torch.zeros(total_strings, dtype=torch.uint8, device=device)

    main_s = torch.cuda.current_stream(device)
    region_pool = [torch.cuda.Stream(device, priority=args.region_priority) for _ in range(max(1, args.region_streams))]
    region_s = region_pool[0]
    walk_pool = [torch.cuda.Stream(device) for _ in range(max(1, args.walk_streams))]
    ev_fork = torch.cuda.Event(enable_timing=True)
    ev_join = torch.cuda.Event(enable_timing=True)
    ev_done = {ex: torch.cuda.Event() for ex in shards}

    def seg(ex):
        a = shards[ex]["first"]
        return a, a + shards[ex]["n"]

    if "," in args.region_launches:
        # explicit group sizes (examples per launch, in layout order)
        sizes_g = [int(t) for t in args.region_launches.split(",")]
        assert sum(sizes_g) == len(layout) and min(sizes_g) > 0, "--region-launches a,b,c: group sizes must add up to the number of examples"
        groups, at = [], 0
        for g_n in sizes_g:
            groups.append(layout[at:at + g_n]); at += g_n
    else:
        n_groups = {"per-example": len(layout), "one": 1}.get(args.region_launches) or max(1, min(len(layout), int(args.region_launches)))
        # groups of consecutive examples with about equal bytes
        groups, acc, cur = [], 0, []
        for ex in layout:
            cur.append(ex)
            acc += shards[ex]["nbytes"]
            if acc >= total_bytes * (len(groups) + 1) / n_groups - 1 and len(groups) < n_groups - 1:
                groups.append(cur); cur = []
        if cur:
            groups.append(cur)
    ev_g0 = [torch.cuda.Event(enable_timing=True) for _ in groups]
    ev_g1 = [torch.cuda.Event(enable_timing=True) for _ in groups]

    def launch_all(where):
        """one pass over the mixed batch: the region pre-pass group by group on the region stream(s), each example's walk on its
        stream as soon as its group's regions are known"""
        ev_fork.record(main_s)
        for rs in region_pool:
            rs.wait_event(ev_fork)
        for k, grp in enumerate(groups):
            rs = region_pool[k % len(region_pool)]
            a, b = seg(grp[0])[0], seg(grp[-1])[1]
            ev_g0[k].record(rs)
            capi.region_scan(bytes_all, off_all[a:b + 1], table[a:b], stream=rs)
            ev_g1[k].record(rs)
            for ex in grp:
                a, b = seg(ex)
                st = where[ex]
                st.wait_event(ev_g1[k])
                if args.exp != "region-only" or not walked[0]:
                    shards[ex]["img"].match_tensors_regions(bytes_all, off_all[a:b + 1], table[a:b], results[a:b], stream=st)
                ev_done[ex].record(st)
        for ex in layout:
            main_s.wait_event(ev_done[ex])
        ev_join.record(main_s)

    # set-up (untimed): one pass with every walk on one stream measures each example's walk time; the walks are then spread
    # over the walk streams
    where = {ex: walk_pool[0] for ex in shards}
    walked = [False]
    launch_all(where)
    torch.cuda.synchronize()
    walked[0] = True
    cost = {ex: shards[ex]["img"].last_kernel_ms(local) for ex in shards}
    # list scheduling with release times: a group's walks may start when its region launch has ended (estimated from this pass's
    # region launches); each walk goes to the stream that can start it first.  Inside a step a walk takes about 1.5 x its time alone.
    ready, t_acc = [], 0.0
    for k in range(len(groups)):
        t_acc += ev_g0[k].elapsed_time(ev_g1[k])
        ready.append(t_acc)
    free_at = [0.0] * len(walk_pool)
    for g_k, grp in enumerate(groups):
        for ex in grp:
            k = min(range(len(walk_pool)), key=lambda j: max(free_at[j], ready[g_k]))
            free_at[k] = max(free_at[k], ready[g_k]) + 1.5 * cost[ex]
            where[ex] = walk_pool[k]
    order = layout

    kernel_ms = {ex: [] for ex in shards}
    span_ms, region_all_ms = [], []
    counts = [total_strings] * world

    def step(record):
        launch_all(where)
        full = sharding.gather_results(results, counts, dist, rank, world, comm_device=comm_dev) if dist else None
        if dist is None:
            full = sharding.pack_bitmap(results)                 # the bitmap a gather would send
        if record:
            ev_join.synchronize()
            span_ms.append(ev_fork.elapsed_time(ev_join))
            for ex, sh in shards.items():
                kernel_ms[ex].append(sh["img"].last_kernel_ms(local))
            region_all_ms.append(sum(ev_g0[k].elapsed_time(ev_g1[k]) for k in range(len(groups))))
        return full

    def fence():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gathered = step(True)
    fence()
    dt = time.perf_counter() - t0
    bytes_by_rank = [total_bytes]
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        bl = [torch.zeros(1, dtype=torch.int64, device=comm_dev) for _ in range(world)]
        dist.all_gather(bl, torch.tensor([total_bytes], dtype=torch.int64, device=comm_dev))
        bytes_by_rank = [int(b.item()) for b in bl]

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        job_bytes = sum(bytes_by_rank)
        value = job_bytes * args.steps / dt / 1e9
        if dist:                                              # rank 0 holds every rank's answers, in rank order
            assert gathered is not None and gathered.numel() == total_strings * world
            assert torch.equal(gathered[:total_strings].to(results.device), (results == 1).to(torch.uint8))
        # roofline: algorithmic bytes of a step (1 B per input character + 8 B offset + 1 B result per string) over the device time
        # of the step's kernels = the span from the fork event (before the first region launch) to the join event (after the last
        # walk), HIP events on the streams the kernels run on
        alg = total_bytes + 9 * total_strings
        kern_s = float(np.mean(span_ms)) * 1e-3
        achieved = alg / kern_s / 1e9
        reg_total_ms = float(np.mean(region_all_ms))
        walk_total_ms = float(sum(np.mean(kernel_ms[ex]) for ex in shards))
        per_ex = {}
        for ex, sh in shards.items():
            a, b = seg(ex)
            per_ex[str(ex)] = {"bytes": sh["nbytes"], "walk_ms": float(np.mean(kernel_ms[ex])),
                               "accepted": int((results[a:b] == 1).sum().item())}
        traffic = None
        try:
            with open(TRAFFIC_FILE) as f:
                tj = json.load(f)
            wl = tj["workload"]
            if (wl["strings_per_example"], wl["min_len"], wl["max_len"]) == (n_per, args.min_len, args.max_len):
                traffic = tj["hbm_bytes_per_step"]
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "input GB/s (chars matched/sec) on 10-example attack corpus",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "ranks_seen": dist.get_world_size() if dist else 1, "backend": (dist.get_backend() if dist else None),
            "bytes_by_rank": bytes_by_rank,
            "config": {"workload": "10 README MFA examples (plain mode) as ONE mixed batch, %d pumped attack strings per example per GPU, "
                                   "pump size log-uniform [%d, %d], alternating with/without suffix "
                                   "(BASELINE configs[3] shard: 10M strings over 8 GPUs)" % (n_per, args.min_len, args.max_len),
                       "strings_per_example": n_per, "min_len": args.min_len, "max_len": args.max_len, "strings_per_gpu": total_strings,
                       "bytes_per_gpu": total_bytes, "parallelism": "dp%d" % world,
                       "exchange": "gather of the result bitmap to rank 0" + (" (%s)" % dist.get_backend() if dist else " (single rank: none)")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "region_scan_kernel (%s) + mfa_jit_kernel (10 launches on %d streams, costliest first); duration = fork-to-join span of a step" % (
                             "%d launch(es) over groups of consecutive examples %s" % (len(groups), groups), len(walk_pool)),
                         "algorithmic_bytes_per_step": alg, "kernel_seconds_per_step": kern_s,
                         # the kernel that reads the bytes: its own launches, timed with HIP events on its stream
                         "region_scan_kernel": {"launches_per_step": len(groups), "ms_per_step": reg_total_ms,
                                                "ms_per_launch": reg_total_ms / len(groups),
                                                "achieved": total_bytes / (reg_total_ms * 1e-3) / 1e9, "frac": total_bytes / (reg_total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                "note": "runs beside the walk kernels of earlier examples"},
                         "mfa_jit_kernel": {"launches_per_step": len(shards), "ms_sum_per_step": walk_total_ms,
                                            "dispatch_ms_mean": walk_total_ms / len(shards)}},
            "order": order, "per_example": per_ex,
        }
        if not args.no_cpu_baseline and world == 1:
            gpu_res = {ex: results[sh["first"]:sh["first"] + PARITY_N].cpu().numpy() for ex, sh in shards.items()}
            out["cpu_baseline"] = cpu_baseline(corpus, shards, gpu_res)
        if not args.no_secondary and world == 1:
            del bytes_all, off_all, table
            torch.cuda.empty_cache()
            sec = [secondary_dfa(device, capi), secondary_config3(device, capi), secondary_64k_all_examples(device, capi, corpus)]
            sec += secondary_no_regions(device, capi, corpus)
            sec += secondary_config5(device, capi, corpus)
            out["secondary"] = sec
        print(json.dumps(out))
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_last.json"), "w") as f:
                f.write(json.dumps(out) + "\n")
        except OSError:
            pass
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
