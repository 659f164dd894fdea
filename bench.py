#!/usr/bin/env python3
"""Benchmark of the match path: input GB/s on the 10-example attack corpus (BASELINE.json metric).

One step = one pass of the hot path over this rank's shard of the mixed corpus (BASELINE configs[3]): ONE batch
(one byte buffer, one offset array) that holds the strings of all ten README examples, example after example,
matched by ONE library call (mfa_match_mixed: the region pre-pass over groups of strings and the walks of every
group beside the next group's pre-pass, scheduled inside the library); then the result bitmap of the shard is
gathered to rank 0 (RCCL when N > 1).
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
KERNEL_NAMES = {1: "walk_kernel (table-driven)", 2: "mfa_jit_kernel (generated for the automaton)", 3: "dfa_tiled_kernel"}
PARITY_FRACTION = 0.01         # seeded sample of every example's shard (no length cap) re-checked on the CPU after the timed region
BASELINE_N = 48                # first strings of every example kept on the host for the reference-build baseline
GOLDEN = os.path.join(ROOT, "tests", "golden")
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r04_traffic.json")      # PMC counters of one headline step (tools/profile_round.sh), stamped with the kernel sources' hash


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
    # preflight, before anything is forked: the native library loads and exports what this script calls; with the generated
    # kernels asked for, every example's code object is in the cache (host-only work: N ranks must not meet on a cold compiler)
    sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd"))
    from mfa_amd import capi, corpus
    capi.lib()
    if args.engine == "jit" or env.get("MFA_WALK") == "jit":
        for ex in corpus.EXAMPLES:
            capi.Image(load_blob("ex%d_plain" % ex)).specialize()
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
    # wait for every rank; one that fails or does not finish in time takes its siblings down (exact pids, never a pattern)
    deadline = time.monotonic() + float(env.get("MFA_BENCH_RANK_TIMEOUT", "1500"))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            try:
                code = p.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            pending.remove(p)
            rc = max(rc, abs(code))
        if pending and (rc != 0 or time.monotonic() > deadline):
            if rc == 0:
                sys.stderr.write("bench.py: a rank did not finish in time\n")
                rc = 124
            for p in pending:
                p.kill()
            for p in pending:
                p.wait()
            break
    return rc


def host_cores(limit=32):
    """this process's CPU share (the GPU box gives one GPU's share of the host, not all of its cores), `limit` at most"""
    return max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), limit))


def hbm_preflight(n_per, min_len, max_len, world, rank, device_index):
    """What a rank's shard needs in HBM against what its device has free, printed before anything is allocated (stderr, every rank):
    the batch, its offsets and results, the region table (128 B per string), and the generator's temporaries (8 bytes-sized int64
    vectors over a 128 MiB chunk).  Returns (need, free) in bytes."""
    import math
    import torch
    mean_len = (max_len - min_len) / max(math.log(max_len / max(min_len, 1)), 1e-9) if max_len > min_len else max_len      # log-uniform pump sizes
    strings = 10 * n_per
    need = int(strings * (mean_len + 8 + 1 + 128) + 9 * (1 << 27) * 8)
    free, total = torch.cuda.mem_get_info(device_index)
    sys.stderr.write("bench.py: rank %d/%d device %d: shard needs about %.1f GB of HBM (batch %.1f GB + tables and temporaries), %.1f GB free of %.1f GB\n"
                     % (rank, world, device_index, need / 1e9, strings * mean_len / 1e9, free / 1e9, total / 1e9))
    if need > free:
        sys.stderr.write("bench.py: rank %d: NOT ENOUGH free HBM for this shard\n" % rank)
    return need, free


def device_facts(device_index):
    """what the runtime says about rank 0's device (boxes of one pool differ by 8 % on the same build: DESIGN.md section 4.3)"""
    import torch
    p = torch.cuda.get_device_properties(device_index)
    facts = {"name": p.name, "arch": getattr(p, "gcnArchName", None), "compute_units": p.multi_processor_count, "hbm_bytes": p.total_memory}
    for key in ("clock_rate", "memory_clock_rate", "memory_bus_width", "L2_cache_size"):
        if hasattr(p, key):
            facts[key] = getattr(p, key)
    return facts


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
        # the CPU restatement timed on the same sample (1 thread, then every core)
        checked, mism, port_bytes, port_sec = 0, 0, 0, 0.0
        all_cores = None
        if os.path.exists(cli):
            jobs = []
            for ex, sh in shards.items():
                sample = [s for s in sh["sample"] if len(s) <= 32768]      # (timing sample; the parity sample is parity_sample())
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
            cores = host_cores()
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


# ---- parity at scale: a seeded sample of the batch against the CPU restatement, every host core ----------------------
def _oracle_match(args):
    cli, blob_path, text = args
    p = subprocess.run([cli, "match", blob_path], input=text, capture_output=True)
    if p.returncode != 0:
        return None
    return [int(x) for x in p.stdout.split()]


def parity_sample(jobs, cores=None):
    """jobs: list of (blob, strings, gpu answers).  The CPU restatement (oracle/oracle_cli, test infrastructure: the checker,
    never the thing measured) on every string, one process per core; returns {strings, bytes, max_len, mismatches}."""
    cli = os.path.join(ROOT, "oracle", "oracle_cli")
    if not os.path.exists(cli):
        return None
    cores = cores or host_cores()
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    n_str = n_bytes = max_len = mism = 0
    with tempfile.TemporaryDirectory() as tmp:
        work, owners = [], []
        for j, (blob, strings, got) in enumerate(jobs):
            path = os.path.join(tmp, "j%d.blob" % j)
            with open(path, "wb") as f:
                f.write(blob)
            # longest first, dealt round-robin: the cores finish together
            order = sorted(range(len(strings)), key=lambda k: -len(strings[k]))
            for c in range(cores):
                idx = order[c::cores]
                if idx:
                    work.append((cli, path, b"".join(strings[k] + b"\n" for k in idx)))
                    owners.append((j, idx))
            n_str += len(strings); n_bytes += sum(len(x) for x in strings); max_len = max([max_len] + [len(x) for x in strings])
        with ThreadPoolExecutor(max_workers=cores) as pool:
            outs = list(pool.map(_oracle_match, work))
        for (j, idx), out in zip(owners, outs):
            got = jobs[j][2]
            if out is None or len(out) != len(idx):
                mism += len(idx)
                continue
            mism += sum(1 for k, w in zip(idx, out) if int(got[k]) != w)
    return {"strings": n_str, "bytes": n_bytes, "max_len": max_len, "mismatches": mism, "cores": cores,
            "seconds": time.perf_counter() - t0, "checker": "oracle/oracle_cli (CPU restatement, pinned against the reference's own answers by tests/golden)"}


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


def secondary_config3(device, capi, n_strings=1 << 20, length=65536):
    """BASELINE.json configs[2] at its full size (roofline variant of SURVEY section 8d): example 1, 1M strings of exactly 64 KiB
    (68.7 GB), mix by j mod 4: a^(L-1) b, a^L, a^L with one byte at a seeded position set to b, i.i.d. {a: 0.99, b: 0.01}."""
    import numpy as np
    import torch
    blob = load_blob("ex1_plain")
    img = capi.Image(blob)
    g = torch.Generator(device=device); g.manual_seed(0x5EED0003)
    flat = torch.empty(n_strings * length + 64, dtype=torch.uint8, device=device)
    flat.fill_(ord("a"))
    flat[-64:] = 0
    data = flat[:n_strings * length].view(n_strings, length)
    data[0::4, -1] = ord("b")
    rows = torch.arange(2, n_strings, 4, device=device)
    data[rows, torch.randint(0, length, (rows.numel(),), generator=g, device=device)] = ord("b")
    noise = data.view(n_strings // 4, 4, length)[:, 3, :]          # the random rows: a strided view, filled in place, in chunks
    for lo in range(0, n_strings // 4, 4096):
        hi = min(n_strings // 4, lo + 4096)
        mask = torch.rand((hi - lo, length), generator=g, device=device) < 0.01
        noise[lo:hi].masked_fill_(mask, ord("b"))
        del mask
    off = torch.arange(0, (n_strings + 1) * length, length, dtype=torch.int64, device=device)
    res = torch.empty(n_strings, dtype=torch.uint8, device=device)
    t, tr = _timed(img, flat, off, res, device, reps=2)
    nbytes = n_strings * length
    # the same batch through mfa_match_mixed (one image): groups of strings, the walks of a group beside the region pass of the next
    mx = capi.Mixed([img])
    res_m = torch.empty(n_strings, dtype=torch.uint8, device=device)
    spans = []
    for _ in range(3):
        mx.match_tensors(flat, off, [0, n_strings], res_m); torch.cuda.synchronize()
        spans.append(mx.last_ms(device.index or 0))
    span, span_region = float(np.mean([x[1] for x in spans[1:]])), float(np.mean([x[0] for x in spans[1:]]))
    # (several groups: the mixed call walks with the table engine unless the generated kernels are asked for; one group: it is the single-automaton call)
    mixed_kernel = KERNEL_NAMES[2] if os.environ.get("MFA_WALK") == "jit" else (KERNEL_NAMES[1] if mx.last_launches(device.index or 0)["groups"] > 1 else KERNEL_NAMES.get(img.info()["last_kernel"], "?"))
    same = bool(torch.equal(res, res_m))
    mx.close()
    img.match_tensors(flat, off, res)          # (the per-image engine again: the kernel name reported below)
    del res_m
    # strings with more periodic stretches than a table row holds (text made of hundreds of medium runs, here the a/b noise strings):
    # the pass reads them to their end all the same and keeps the longest stretches
    tab = capi.region_scan(flat, off)
    over = int(((tab[:, 0] & capi.REGION_OVERFLOW) != 0).sum().item())
    del tab
    # a^L is accepted (SURVEY section 8c anchors: aa, aaa, aaaa, aaaaaaaa -> 1), every string containing a b is not
    ok = bool(res[1::4].all().item()) and not bool(res[0::4].any().item()) and not bool(res[2::4].any().item()) and not bool(res[3::4].any().item())
    # parity: a seeded 0.1 % of the strings, full length, against the CPU restatement
    rng = np.random.Generator(np.random.Philox(0x5EED0013))
    idx = np.sort(rng.choice(n_strings, size=max(8, n_strings // 1024), replace=False))
    strings = [bytes(data[int(k)].cpu().numpy().tobytes()) for k in idx]
    par = parity_sample([(blob, strings, res[torch.from_numpy(idx).to(device)].cpu().numpy())])
    return {"workload": "configs[2]: ({a*}:1&1)*, %d strings of exactly %d bytes (%.1f GB), 4-way attack mix" % (n_strings, length, nbytes / 1e9),
            "kernel": "region_scan_kernel + " + KERNEL_NAMES.get(img.info()["last_kernel"], "?"), "region_ms": tr, "walk_ms": t,
            "GB/s_on_sum_of_lengths": nbytes / ((t + tr) * 1e-3) / 1e9,
            "mixed_call": {"kernel": "mfa_match_mixed: region_scan_kernel + " + mixed_kernel, "span_ms": span, "region_ms": span_region, "GB/s_on_sum_of_lengths": nbytes / (span * 1e-3) / 1e9,
                           "frac_of_hbm_peak_on_touched_bytes": nbytes / (span * 1e-3) / 1e9 / HBM_PEAK_GBS, "results_equal": same},
            # the walk stops at the first empty state set (mfa.cpp:224-225), but the region pass has read every byte by then
            "touched_bytes": nbytes,
            "touched_by": "region_scan_kernel reads every byte of every string (%d strings, %.1f %%, have more stretches than a table row holds: "
                          "the longest are kept); the walk reads only the bytes of the steps it executes" % (over, 100.0 * over / n_strings),
            "frac_of_hbm_peak_on_touched_bytes": nbytes / ((t + tr) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "region_pass_GB/s": nbytes / (tr * 1e-3) / 1e9 if tr > 0 else None,
            "results_as_expected": ok, "parity_sample": par}


def secondary_config3_loguniform(device, capi, corpus, n_strings=1 << 20, lo=1024, hi=65536):
    """BASELINE.json configs[2], the SECONDARY variant of SURVEY section 8d (config 3): example 1, 1M strings, the same 4-way mix by
    j mod 4 (a^(L-1) b, a^L, a^L with one byte at a seeded position set to b, i.i.d. {a: 0.99, b: 0.01}) with lengths log-uniform in
    [1 KiB, 64 KiB], seed 0x5EED0003; generated on the device.  Its own parity sample (0.1 % of the strings, full length)."""
    import numpy as np
    import torch
    blob = load_blob("ex1_plain")
    img = capi.Image(blob)
    lens = corpus.pump_sizes(n_strings, 0x5EED0003, lo, hi)
    off_h = np.zeros(n_strings + 1, dtype=np.int64)
    np.cumsum(lens, out=off_h[1:])
    total = int(off_h[-1])
    g = torch.Generator(device=device); g.manual_seed(0x5EED0003)
    flat = torch.empty(total + 64, dtype=torch.uint8, device=device)
    flat.fill_(ord("a"))
    flat[-64:] = 0
    off = torch.from_numpy(off_h).to(device)
    j = torch.arange(n_strings, device=device)
    flat[off[1:][j % 4 == 0] - 1] = ord("b")                                           # kind 0: a^(L-1) b
    k2 = j[j % 4 == 2]
    pos = (torch.rand(k2.numel(), generator=g, device=device, dtype=torch.float64) * torch.from_numpy(lens).to(device)[k2]).long()
    flat[off[:-1][k2] + pos] = ord("b")                                                # kind 2: one b somewhere
    chunk = 1 << 27                                                                    # kind 3: noise, over the strings' own bytes only
    for a in range(0, total, chunk):
        b = min(total, a + chunk)
        p = torch.arange(a, b, device=device)
        sid = torch.searchsorted(off, p, right=True) - 1
        m = ((sid & 3) == 3) & (torch.rand(b - a, generator=g, device=device) < 0.01)
        flat[a:b].masked_fill_(m, ord("b"))
        del p, sid, m
    res = torch.empty(n_strings, dtype=torch.uint8, device=device)
    t, tr = _timed(img, flat, off, res, device, reps=2)
    mx = capi.Mixed([img])
    res_m = torch.empty(n_strings, dtype=torch.uint8, device=device)
    spans = []
    for _ in range(3):
        mx.match_tensors(flat, off, [0, n_strings], res_m); torch.cuda.synchronize()
        spans.append(mx.last_ms(device.index or 0))
    span = float(np.mean([x[1] for x in spans[1:]]))
    same = bool(torch.equal(res, res_m))
    mx.close()
    ok = bool(res[1::4].all().item()) and not bool(res[0::4].any().item()) and not bool(res[2::4].any().item()) and not bool(res[3::4].any().item())
    rng = np.random.Generator(np.random.Philox(0x5EED0015))
    idx = np.sort(rng.choice(n_strings, size=max(8, n_strings // 1024), replace=False))
    strings = [bytes(flat[int(off_h[k]):int(off_h[k + 1])].cpu().numpy().tobytes()) for k in idx]
    par = parity_sample([(blob, strings, res[torch.from_numpy(idx).to(device)].cpu().numpy())])
    return {"workload": "configs[2], secondary variant: ({a*}:1&1)*, %d strings, lengths log-uniform [%d, %d] (%.1f GB), 4-way attack mix" % (n_strings, lo, hi, total / 1e9),
            "kernel": "region_scan_kernel + " + KERNEL_NAMES.get(img.info()["last_kernel"], "?"), "region_ms": tr, "walk_ms": t,
            "GB/s": total / ((t + tr) * 1e-3) / 1e9, "touched_bytes": total,
            "frac_of_hbm_peak_on_touched_bytes": total / ((t + tr) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "mixed_call": {"span_ms": span, "GB/s": total / (span * 1e-3) / 1e9, "frac_of_hbm_peak_on_touched_bytes": total / (span * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "results_equal": same},
            "closed_form_check": ok, "accepted": int(res.sum().item()), "parity_sample": par}


def secondary_64k_all_examples(device, capi, corpus, n_strings=16384):
    """north_star: "at 64 KiB strings" -- all ten examples, every string pumped to 64 KiB (pump size 65536), alternating
    with / without suffix: ONE mixed batch through the same mfa_match_mixed call as the headline."""
    import numpy as np
    import torch
    layout = [2, 5, 3, 8, 9, 10, 6, 4, 1, 7]
    parts_b, parts_o, seg, pos_b, images, blobs = [], [], [0], 0, [], []
    sizes = np.full(n_strings, 65536, dtype=np.int64)
    ws = (np.arange(n_strings) % 2) == 0
    for ex in layout:
        b, o = corpus.device_batch(ex, sizes, ws, device)
        nb = int(o[-1].item())
        parts_b.append(b[:nb]); parts_o.append(o[:-1] + pos_b); pos_b += nb; seg.append(seg[-1] + n_strings)
        blobs.append(load_blob("ex%d_plain" % ex))
        images.append(capi.Image(blobs[-1]))
        del b, o
    bytes_all = torch.cat(parts_b + [torch.zeros(64, dtype=torch.uint8, device=device)])
    off_all = torch.cat(parts_o + [torch.tensor([pos_b], dtype=torch.int64, device=device)])
    del parts_b, parts_o
    mx = capi.Mixed(images)
    res = torch.zeros(seg[-1], dtype=torch.uint8, device=device)
    wall, spans, regs = [], [], []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mx.match_tensors(bytes_all, off_all, seg, res); torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
        r_ms, sp_ms = mx.last_ms(device.index or 0)
        spans.append(sp_ms); regs.append(r_ms)
    span = float(np.mean(spans[2:]))
    # parity: 1 % of every example's strings (they are all of the full size) against the CPU restatement
    rng = np.random.Generator(np.random.Philox(0x5EED0014))
    jobs = []
    for k, ex in enumerate(layout):
        idx = np.sort(rng.choice(n_strings, size=max(4, n_strings // 100), replace=False))
        jobs.append((blobs[k], corpus.host_strings(ex, sizes[idx], ws[idx]), res[seg[k] + torch.from_numpy(idx).to(device)].cpu().numpy()))
    out = {"workload": "north_star size: 10 examples as one mixed batch, %d strings each, pump size exactly 65536, alternating with/without suffix" % n_strings,
           "kernel": "mfa_match_mixed: region_scan_kernel + " + KERNEL_NAMES.get(images[0].info()["last_kernel"], "walk_kernel"),
           "bytes": pos_b, "span_ms": span, "region_ms": float(np.mean(regs[2:])), "wall_ms": float(np.mean(wall[2:])),
           "GB/s": pos_b / (span * 1e-3) / 1e9, "touched_bytes": pos_b, "frac_of_hbm_peak_on_touched_bytes": pos_b / (span * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "accepted": int((res == 1).sum().item()), "parity_sample": parity_sample(jobs)}
    mx.close()
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
                      "kernel": "region_scan_kernel + " + KERNEL_NAMES.get(img.info()["last_kernel"], "?"), "region_ms": tr, "walk_ms": t, "GB/s": nb / ((t + tr) * 1e-3) / 1e9,
                      "char_steps_per_s": nb / ((t + tr) * 1e-3), "accepted": int(res.sum().item()),
                      "parity_oracle_sample": _oracle_sample(blob, short, got)})
        # the same batch through mfa_match_mixed (the table engine: strings without periodic stretches are handed to walk_lean_kernel -- the
        # plain step only, twice the waves per SIMD -- whatever the automaton, no generated kernel needed)
        prev_engine = os.environ.get("MFA_WALK")
        os.environ["MFA_WALK"] = "table"                 # (a mixed object of ONE automaton and one group is the single-automaton call: ask for its table engine)
        mx = capi.Mixed([img])
        res_m = torch.empty(n_strings, dtype=torch.uint8, device=device)
        spans = []
        for _ in range(3):
            mx.match_tensors(d_b, d_o, [0, n_strings], res_m, total_bytes=nb); torch.cuda.synchronize()
            spans.append(mx.last_ms(device.index or 0)[1])
        span = float(np.mean(spans[1:]))
        assert img.info()["last_kernel"] == capi.KERNEL_WALK
        if prev_engine is None:
            del os.environ["MFA_WALK"]
        else:
            os.environ["MFA_WALK"] = prev_engine
        lines.append({"workload": "non-periodic text, example %d, the same batch through mfa_match_mixed with the table engine (no generated kernel)" % ex,
                      "kernel": "mfa_match_mixed: region_scan_kernel + walk_kernel (table-driven: hands the strings on) + walk_lean_kernel (walks them)", "span_ms": span, "GB/s": nb / (span * 1e-3) / 1e9,
                      "char_steps_per_s": nb / (span * 1e-3), "results_equal": bool(torch.equal(res, res_m)), "launches": mx.last_launches(device.index or 0)})
        mx.close()
        del d_b, d_o, res, res_m
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
    lines.append({"workload": "example 1, %d attack strings of 4 KiB, NO region table (every step executed)" % n1, "kernel": KERNEL_NAMES.get(img.info()["last_kernel"], "?"),
                  "walk_ms": t, "GB/s": nb / (t * 1e-3) / 1e9, "char_steps_per_s": nb / (t * 1e-3), "accepted": int(res.sum().item())})
    return lines


def secondary_config5(device, capi, corpus, n_strings=125000):
    """BASELINE.json configs[4]: reversed MFAs (`-reverse`, is_reversed = 1) of the nondeterministic examples 3, 6, 8 on
    pump-only strings (full walk) and pump+suffix strings (early exit), reported separately; 1 % of the strings of every line,
    whatever their length, against the CPU restatement."""
    import numpy as np
    import torch
    out = []
    for ex in (3, 6, 8):
        blob = load_blob("ex%d_reverse" % ex)
        img = capi.Image(blob)
        n = n_strings
        for tag, suffix in (("pump only", False), ("pump + suffix", True)):
            sizes = corpus.pump_sizes(n, 0x5EED0005 + ex, 1024, 65536)
            ws = np.full(n, suffix)
            flat, off = corpus.device_batch(ex, sizes, ws, device)
            res = torch.empty(n, dtype=torch.uint8, device=device)
            t, tr = _timed(img, flat, off, res, device, reps=2)
            nbytes = int(off[-1].item())
            # the same batch through mfa_match_mixed (one image): the walks of a group of strings beside the region pass of the next
            mx = capi.Mixed([img])
            res_m = torch.empty(n, dtype=torch.uint8, device=device)
            spans = []
            for _ in range(3):
                mx.match_tensors(flat, off, [0, n], res_m); torch.cuda.synchronize()
                spans.append(mx.last_ms(device.index or 0)[1])
            span = float(np.mean(spans[1:]))
            mixed = {"kernel": "mfa_match_mixed: region_scan_kernel + " + KERNEL_NAMES.get(img.info()["last_kernel"], "?"), "span_ms": span,
                     "GB/s": nbytes / (span * 1e-3) / 1e9, "frac_of_hbm_peak_on_touched_bytes": nbytes / (span * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "results_equal": bool(torch.equal(res, res_m))}
            mx.close()
            del res_m
            img.match_tensors(flat, off, res)          # (the per-image engine again: the kernel name reported below)
            rng = np.random.Generator(np.random.Philox(0x5EED0015 + ex))
            idx = np.sort(rng.choice(n, size=max(4, n // 100), replace=False))
            par = parity_sample([(blob, corpus.host_strings(ex, sizes[idx], ws[idx]), res[torch.from_numpy(idx).to(device)].cpu().numpy())])
            out.append({"workload": "configs[4]: example %d -reverse, %d strings, %s" % (ex, n, tag),
                        "kernel": "region_scan_kernel + " + KERNEL_NAMES.get(img.info()["last_kernel"], "?"),
                        "region_ms": tr, "walk_ms": t, "GB/s": nbytes / ((t + tr) * 1e-3) / 1e9,
                        "touched_bytes": nbytes, "touched_by": "region pass reads every byte; the walk may exit early",
                        "frac_of_hbm_peak_on_touched_bytes": nbytes / ((t + tr) * 1e-3) / 1e9 / HBM_PEAK_GBS, "mixed_call": mixed,
                        "accepted": int(res.sum().item()), "parity_sample": par})
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
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank generates --strings-per-example strings of each example (fixed work per GPU); strong: ONE batch of "
                         "10 x --strings-per-example strings is cut into --gpus ranges of about equal bytes (sharding.partition_by_bytes) and "
                         "every rank generates and matches its range only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--engine", choices=["auto", "table", "jit"], default="auto",
                    help="walk kernels behind mfa_match_mixed: the table-driven walk (one launch per group of strings, any automaton per lane) or the "
                         "kernels generated per automaton (one launch per example); auto = the library's choice (MFA_WALK)")
    ap.add_argument("--cuts", default="", help="development: MFA_MIXED_CUTS, the fractions of the batch at which the region pre-pass is cut into groups")
    ap.add_argument("--order", default="", help="comma-separated order of the examples' segments in the batch (default: costliest walk first)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch
    from mfa_amd import capi, corpus, sharding
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if args.engine != "auto":
        os.environ["MFA_WALK"] = args.engine
    if args.cuts:
        os.environ["MFA_MIXED_CUTS"] = args.cuts

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
    # MFA_BENCH_FORCE_DIST=1: a single rank still goes through torch.distributed (backend nccl = RCCL: communicator on the device,
    # all_gather / gather / all_reduce / barrier on device tensors) -- what a 1-GPU box can execute of the N > 1 path
    force_dist = os.environ.get("MFA_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
        if "MASTER_PORT" not in os.environ:
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            s_.close()
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if world > 1 or force_dist:
        hbm_preflight(args.strings_per_example, args.min_len, args.max_len, world, rank, local)
    comm_dev = torch.device("cpu") if rehearse else device

    # ---- this rank's shard: ONE mixed batch, resident in HBM ------------------------------------------
    # segment order in the batch: costliest walk first (measured once on MI355X; any order is correct)
    layout = [int(x) for x in args.order.split(",")] if args.order else [2, 5, 3, 8, 9, 10, 6, 4, 1, 7]
    assert sorted(layout) == sorted(corpus.EXAMPLES)
    n_per = args.strings_per_example
    # what this rank holds of every example: (first string of the example's stream, count)
    if args.scaling == "weak" or world == 1:
        own = {ex: (0, n_per) for ex in layout}
        seed_of = lambda ex: 0x5EED0004 + 1000 * rank + ex
        cuts = None
    else:
        # strong scaling: the job's ONE batch (every example's n_per strings, seeds of rank 0) cut into `world` ranges of about equal
        # bytes; every rank derives the same cut points from the lengths alone (no data is generated for them) and generates its
        # own range only
        lens_all = np.concatenate([corpus.layout(ex, corpus.pump_sizes(n_per, 0x5EED0004 + ex, args.min_len, args.max_len), (np.arange(n_per) % 2) == 0)["lens"]
                                   for ex in layout])
        off_job = np.zeros(len(lens_all) + 1, dtype=np.int64)
        np.cumsum(lens_all, out=off_job[1:])
        cuts = sharding.partition_by_bytes(off_job, world)
        lo, hi = int(cuts[rank]), int(cuts[rank + 1])
        own = {}
        for k, ex in enumerate(layout):
            a, b = max(lo, k * n_per), min(hi, (k + 1) * n_per)
            own[ex] = (a - k * n_per, max(0, b - a))
        seed_of = lambda ex: 0x5EED0004 + ex
        del lens_all, off_job
    shards = {}
    parts_b, parts_o, pos_s, pos_b = [], [], 0, 0
    rng_par = np.random.Generator(np.random.Philox(0x5EED0009 + rank))
    for ex in layout:
        first, count = own[ex]
        sizes = corpus.pump_sizes(n_per, seed_of(ex), args.min_len, args.max_len)[first:first + count]
        with_suffix = ((np.arange(n_per) % 2) == 0)[first:first + count]
        blob = load_blob("ex%d_plain" % ex)
        img = capi.Image(blob)
        img.prepare(local)
        nbytes = 0
        if count:
            d_bytes, d_off = corpus.device_batch(ex, sizes, with_suffix, device)
            nbytes = int(d_off[-1].item())
            parts_b.append(d_bytes[:nbytes])
            parts_o.append(d_off[:-1] + pos_b)
            del d_bytes, d_off
        # the parity sample of this example: a seeded 1 % of its strings, whatever their length
        n_par = max(1, int(round(count * PARITY_FRACTION))) if count else 0
        par_idx = np.sort(rng_par.choice(count, size=n_par, replace=False)) if n_par else np.zeros(0, dtype=np.int64)
        shards[ex] = {"img": img, "n": count, "nbytes": nbytes, "blob": blob, "first": pos_s, "par_idx": par_idx,
                      "par_strings": corpus.host_strings(ex, sizes[par_idx], with_suffix[par_idx]) if rank == 0 else [],
                      "sample": corpus.host_strings(ex, sizes[:BASELINE_N], with_suffix[:BASELINE_N]) if rank == 0 else []}
        pos_s += count
        pos_b += nbytes
    total_strings, total_bytes = pos_s, pos_b
    bytes_all = torch.cat(parts_b + [torch.zeros(64, dtype=torch.uint8, device=device)])
    off_all = torch.cat(parts_o + [torch.tensor([total_bytes], dtype=torch.int64, device=device)])
    del parts_b, parts_o
    torch.cuda.empty_cache()
    results = torch.zeros(max(total_strings, 1), dtype=torch.uint8, device=device)[:total_strings]
    seg = [shards[ex]["first"] for ex in layout] + [total_strings]
    mixed = capi.Mixed([shards[ex]["img"] for ex in layout])

    span_ms, region_all_ms = [], []
    counts_t = torch.tensor([total_strings], dtype=torch.int64, device=comm_dev)
    if dist:
        cl = [torch.zeros(1, dtype=torch.int64, device=comm_dev) for _ in range(world)]
        dist.all_gather(cl, counts_t)
        counts = [int(c.item()) for c in cl]
    else:
        counts = [total_strings]

    def step(record):
        """one pass of the hot path: one library call for the whole mixed batch, then the result bitmap goes to rank 0"""
        mixed.match_tensors(bytes_all, off_all, seg, results, total_bytes=total_bytes)
        full = sharding.gather_results(results, counts, dist, rank, world, comm_device=comm_dev) if dist else None
        if dist is None:
            full = sharding.pack_bitmap(results)                 # the bitmap a gather would send
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
    # device times of the timed steps: HIP events the library recorded on its own streams (it keeps those of its last 32 calls), read
    # after the timed region so that no step waits for the host
    for k in range(min(args.steps, 32)):
        r_ms, sp_ms = mixed.last_ms(local, back=k)
        span_ms.append(sp_ms)
        region_all_ms.append(r_ms)
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
            assert gathered is not None and gathered.numel() == sum(counts)
            assert torch.equal(gathered[:total_strings].to(results.device), (results == 1).to(torch.uint8))
        # roofline: algorithmic bytes of a step (1 B per input character + 8 B offset + 1 B result per string) over the device time
        # of the step's kernels = from the first region launch to the end of the last walk, HIP events on the streams the library
        # launches them on (mfa_mixed_last_ms)
        alg = total_bytes + 9 * total_strings
        kern_s = float(np.mean(span_ms)) * 1e-3
        achieved = alg / kern_s / 1e9
        reg_total_ms = float(np.mean(region_all_ms))
        engine = {capi.KERNEL_WALK: "walk_kernel (table-driven, one launch per group of strings and cell count)",
                  capi.KERNEL_SPECIALISED: "mfa_jit_kernel (one launch per example)"}.get(shards[layout[0]]["img"].info()["last_kernel"], "?")
        if os.environ.get("MFA_WALK", "") != "jit":
            engine = "walk_kernel (table-driven, one launch per group of strings and cell count)"
        per_ex = {}
        for ex, sh in shards.items():
            a = sh["first"]
            per_ex[str(ex)] = {"bytes": sh["nbytes"], "strings": sh["n"], "accepted": int((results[a:a + sh["n"]] == 1).sum().item())}
        # HBM traffic of a step: PMC counters collected in a separate run (FETCH_SIZE / WRITE_SIZE passes).  Quoted only for the workload
        # they were collected on AND while the kernel sources are the ones they were collected with; otherwise null, with the reason
        traffic, traffic_note = None, None
        try:
            from mfa_amd import srchash
            with open(TRAFFIC_FILE) as f:
                tj = json.load(f)
            wl = tj["workload"]
            if (wl["strings_per_example"], wl["min_len"], wl["max_len"]) != (n_per, args.min_len, args.max_len):
                traffic_note = "counters in %s are for another workload" % os.path.basename(TRAFFIC_FILE)
            elif tj.get("csrc_sha16") != srchash.kernel_source_hash():
                traffic_note = "counters in %s were collected with other kernel sources (%s, now %s)" % (
                    os.path.basename(TRAFFIC_FILE), tj.get("csrc_sha16"), srchash.kernel_source_hash())
            else:
                traffic = tj["hbm_bytes_per_step"]
                traffic_note = "%s: separate --pmc passes over one step, FETCH_SIZE doubled (gfx950), kernel sources %s" % (os.path.basename(TRAFFIC_FILE), tj["csrc_sha16"])
        except (OSError, KeyError, ValueError) as e:
            traffic_note = "no counters: %s" % e
        # what a step launches, from the library itself (mfa_mixed_last_launches): with the gate ONE region launch over the whole batch and
        # the walks of a group released by its counter; otherwise one region launch per group of strings
        launches = mixed.last_launches(local)
        n_groups, n_region = launches["groups"], launches["region_launches"]
        out = {
            "metric": "input GB/s (chars matched/sec) on 10-example attack corpus",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "ranks_seen": dist.get_world_size() if dist else 1, "backend": (dist.get_backend() if dist else None),
            "bytes_by_rank": bytes_by_rank, "strings_by_rank": counts,
            "device": device_facts(local),
            "config": {"workload": "10 README MFA examples (plain mode) as ONE mixed batch matched by one mfa_match_mixed call per step, %d pumped attack strings per "
                                   "example%s, pump size log-uniform [%d, %d], alternating with/without suffix "
                                   "(BASELINE configs[3]: 10M strings over 8 GPUs)" % (
                                       n_per, " per GPU" if args.scaling == "weak" or world == 1 else " in the whole job, cut into %d ranges of equal bytes" % world,
                                       args.min_len, args.max_len),
                       "strings_per_example": n_per, "min_len": args.min_len, "max_len": args.max_len, "strings_per_gpu": total_strings,
                       "bytes_per_gpu": total_bytes, "parallelism": "dp%d" % world,
                       "exchange": "gather of the result bitmap to rank 0" + (" (%s)" % dist.get_backend() if dist else " (single rank: none)")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": "region_scan_kernel (%s) + %s in %d launches; duration = first region launch to end of the last walk" % (
                             "ONE launch over the batch, the walks of each of %d groups of strings released by a counter the kernel raises" % n_groups if launches["gated"]
                             else "%d launches over groups of strings, on one stream" % n_region, engine, launches["walk_launches"]),
                         "launches_per_step": launches,
                         "algorithmic_bytes_per_step": alg, "kernel_seconds_per_step": kern_s,
                         "kernel_ms_by_step": [round(x, 3) for x in span_ms[::-1]],
                         # the kernel that reads the bytes: its own launches, back to back on their stream, timed with HIP events
                         "region_scan_kernel": {"launches_per_step": n_region, "ms_per_step": reg_total_ms,
                                                "ms_per_launch": reg_total_ms / max(n_region, 1),
                                                "achieved": total_bytes / (reg_total_ms * 1e-3) / 1e9, "frac": total_bytes / (reg_total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                "note": "runs beside the walk kernels of earlier groups"}},
            "order": layout, "per_example": per_ex,
        }
        if world == 1 or args.scaling == "weak":
            # parity at scale: a seeded 1 % of every example's strings, no length cap, against the CPU restatement
            jobs = []
            for ex in layout:
                sh = shards[ex]
                if len(sh["par_idx"]):
                    got = results[sh["first"] + torch.from_numpy(sh["par_idx"]).to(results.device)].cpu().numpy()
                    jobs.append((sh["blob"], sh["par_strings"], got))
            out["parity_sample"] = parity_sample(jobs)
        if not args.no_cpu_baseline and world == 1:
            gpu_res = {ex: results[sh["first"]:sh["first"] + BASELINE_N].cpu().numpy() for ex, sh in shards.items()}
            out["cpu_baseline"] = cpu_baseline(corpus, shards, gpu_res)
            if out["cpu_baseline"] is not None and out.get("parity_sample"):
                out["cpu_baseline"]["parity_restatement"] = {k: out["parity_sample"][k] for k in ("strings", "bytes", "max_len", "mismatches")}
                out["cpu_baseline"]["parity"] = bool(out["cpu_baseline"]["parity"]) and out["parity_sample"]["mismatches"] == 0
        if not args.no_secondary and world == 1:
            mixed.close()
            del bytes_all, off_all
            torch.cuda.empty_cache()
            sec = [secondary_dfa(device, capi), secondary_config3(device, capi), secondary_config3_loguniform(device, capi, corpus),
                   secondary_64k_all_examples(device, capi, corpus)]
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
