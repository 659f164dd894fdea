#!/usr/bin/env python3
"""Benchmark of the match path: input GB/s on the 10-example attack corpus (BASELINE.json metric).

One step = one pass of the hot path over this rank's shard of the corpus: for each of the ten README
examples (plain-mode MFA, as `./diploma -match` compiles them) one `mfa_match_batch` launch over that
example's strings, then the result bitmap of the shard is gathered to rank 0 (RCCL when N > 1).
Per-GPU work is fixed (weak scaling): rank r holds `--strings-per-example` strings of every example,
`prefix + pumped_string(n, pump) [+ suffix]` with n log-uniform in [--min-len, --max-len]
(generator: reference matchers/example_runner.cpp:15-29), generated on the device before the timed
region, so the timed region starts with all inputs resident in HBM.

Launch (N > 1):  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
                 --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mfa_amd import capi, corpus, image, sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
PARITY_N = 48                  # strings per example whose GPU answers are re-checked on the CPU after the timed region
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_blob(name):
    with open(os.path.join(GOLDEN, "images", name + ".dump")) as f:
        return image.blob_from_dump(f.read())


def cpu_baseline(shards, gpu_results, budget_strings=8, cap=32768):
    """Time the reference itself (oracle/_ref/ref_harness: its own sources, built as it builds them, no
    -O flag, canonical allocation-order mode) on a bounded sample of the same workload; falls back to
    our CPU restatement when the reference build is not present."""
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
        if os.path.exists(cli):
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
    if tot_sec <= 0:
        return None
    return {"value": tot_bytes / tot_sec / 1e9, "unit": "GB/s", "cores": 1, "kind": "reference" if use_ref else "port",
            "sample": "%d strings (first <=%d of each example's shard with length <= %d KiB), %d bytes, %.1f s, 1 thread" % (
                n_str, budget_strings, cap // 1024, tot_bytes, tot_sec),
            # the same strings were matched on the GPU in the timed region: the accept counts must agree
            "accepted": acc_cpu, "accepted_gpu": acc_gpu, "parity": acc_cpu == acc_gpu and mism == 0,
            "parity_restatement": {"strings": checked, "mismatches": mism},
            # the CPU restatement (oracle/mfa_oracle.c, -O2, one thread) on the parity sample, process start included
            "restatement": {"value": port_bytes / port_sec / 1e9 if port_sec > 0 else None, "unit": "GB/s", "cores": 1, "kind": "port",
                            "sample": "%d strings, %d bytes, %.2f s" % (checked, port_bytes, port_sec)}}


def secondary_dfa(device, n_strings=1 << 20, length=1024):
    """BASELINE.json configs[1]: Thompson automaton of (a|b)*abb through the API, 1M random 1 KiB strings."""
    img = capi.Image(load_blob("nfa_abb_thompson"))
    g = torch.Generator(device=device); g.manual_seed(0x5EED0002)
    data = torch.randint(0, 2, (n_strings, length), generator=g, device=device, dtype=torch.uint8) + ord("a")
    data[0::2, -3:] = torch.tensor(list(b"abb"), dtype=torch.uint8, device=device)
    flat = torch.cat([data.reshape(-1), torch.zeros(64, dtype=torch.uint8, device=device)])
    off = torch.arange(0, (n_strings + 1) * length, length, dtype=torch.int64, device=device)
    res = torch.empty(n_strings, dtype=torch.uint8, device=device)
    ms = []
    for _ in range(4):
        img.match_tensors(flat, off, res)
        ms.append(img.last_kernel_ms(device.index or 0))
    torch.cuda.synchronize()
    t = float(np.mean(ms[1:]))
    want = (data[:, -3] == ord("a")) & (data[:, -2] == ord("b")) & (data[:, -1] == ord("b"))
    ok = bool(torch.equal(res.bool(), want))
    gbs = n_strings * length / (t * 1e-3) / 1e9
    return {"workload": "(a|b)*abb Thompson NFA (tabulated), %d random %d-byte strings" % (n_strings, length),
            "kernel": "dfa_walk_kernel", "kernel_ms": t, "GB/s": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS,
            "results_exact": ok}


def _timed(img, flat, off, res, device, reps=3):
    ms = []
    for _ in range(reps + 1):
        img.match_tensors(flat, off, res)
        ms.append(img.last_kernel_ms(device.index or 0))
    torch.cuda.synchronize()
    return float(np.mean(ms[1:]))


def secondary_config3(device, n_strings=1 << 17, length=65536):
    """BASELINE.json configs[2] (roofline variant of SURVEY section 8d): example 1, strings of exactly 64 KiB, mix by
    j mod 4: a^(L-1) b, a^L, a^L with one byte at a seeded position set to b, i.i.d. {a: 0.99, b: 0.01}."""
    img = capi.Image(load_blob("ex1_plain"))
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
    flat = torch.cat([data.reshape(-1), torch.zeros(64, dtype=torch.uint8, device=device)])
    del data
    off = torch.arange(0, (n_strings + 1) * length, length, dtype=torch.int64, device=device)
    res = torch.empty(n_strings, dtype=torch.uint8, device=device)
    t = _timed(img, flat, off, res, device)
    gbs = n_strings * length / (t * 1e-3) / 1e9
    # a^L is accepted (SURVEY section 8c anchors: aa, aaa, aaaa, aaaaaaaa -> 1), every string containing a b is not
    ok = bool(res[1::4].all().item()) and not bool(res[0::4].any().item()) and not bool(res[2::4].any().item())
    return {"workload": "configs[2]: ({a*}:1&1)*, %d strings of exactly %d bytes, 4-way attack mix" % (n_strings, length),
            "kernel": "mfa_jit_kernel", "kernel_ms": t, "GB/s": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "results_as_expected": ok}


def secondary_config5(device, n_strings=125000):
    """BASELINE.json configs[4]: reversed MFAs (`-reverse`, is_reversed = 1) of the nondeterministic examples 3, 6, 8 on
    pump-only strings (full walk) and pump+suffix strings (early exit), reported separately.  Images are the reference's
    own `-reverse` automata (tests/golden/images): this build's front-end has no BNF rewriter yet."""
    out = []
    for ex in (3, 6, 8):
        img = capi.Image(load_blob("ex%d_reverse" % ex))
        n = n_strings if ex != 8 else n_strings // 5             # ex. 8 -reverse has 77 nodes: slot sets in LDS, 16 strings per wave
        for tag, suffix in (("pump only", False), ("pump + suffix", True)):
            sizes = corpus.pump_sizes(n, 0x5EED0005 + ex, 1024, 65536)
            flat, off = corpus.device_batch(ex, sizes, np.full(n, suffix), device)
            res = torch.empty(n, dtype=torch.uint8, device=device)
            t = _timed(img, flat, off, res, device, reps=2)
            nbytes = int(off[-1].item())
            out.append({"workload": "configs[4]: example %d -reverse, %d strings, %s" % (ex, n, tag),
                        "kernel": {capi.KERNEL_GENERIC: "mfa_walk_kernel", capi.KERNEL_SPECIALISED: "mfa_jit_kernel"}[img.info()["last_kernel"]],
                        "kernel_ms": t, "GB/s": nbytes / (t * 1e-3) / 1e9, "accepted": int(res.sum().item())})
            del flat, off, res
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--strings-per-example", type=int, default=125000,
                    help="strings of each example per GPU (125000 x 10 examples x 8 GPUs = the 10M-string batch)")
    ap.add_argument("--min-len", type=int, default=1024)
    ap.add_argument("--max-len", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the ten launches of a step are spread over (longest-processing-time-first); 0 = try 3, 4 and 6 "
                         "during set-up and keep the fastest; "
                         "the runtime maps streams onto four hardware queues, more streams gain nothing")
    ap.add_argument("--sequential", dest="concurrent", action="store_false",
                    help="launch the ten examples one after another on one stream instead of on ten streams")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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

    # ---- this rank's shard, resident in HBM ---------------------------------------------------------
    shards = {}
    n_per = args.strings_per_example
    total_bytes, total_strings = 0, 0
    for ex in sorted(corpus.EXAMPLES):
        seed = 0x5EED0004 + 1000 * rank + ex
        sizes = corpus.pump_sizes(n_per, seed, args.min_len, args.max_len)
        with_suffix = (np.arange(n_per) % 2) == 0
        d_bytes, d_off = corpus.device_batch(ex, sizes, with_suffix, device)
        blob = load_blob("ex%d_plain" % ex)
        img = capi.Image(blob)
        img.prepare(local)
        nbytes = int(d_off[-1].item())
        shards[ex] = {"img": img, "bytes": d_bytes, "off": d_off, "n": n_per, "nbytes": nbytes, "blob": blob,
                      "sample": corpus.host_strings(ex, sizes[:PARITY_N], with_suffix[:PARITY_N]) if rank == 0 else []}
        total_bytes += nbytes
        total_strings += n_per
    results = torch.zeros(total_strings, dtype=torch.uint8, device=device)
    n_bitmap = (total_strings + 7) // 8
    comm_dev = torch.device("cpu") if rehearse else device
    gathered = [torch.empty(n_bitmap, dtype=torch.uint8, device=comm_dev) for _ in range(world)] if (dist and rank == 0) else None
    kernel_ms = {ex: [] for ex in shards}
    span_ms = []
    # one HIP stream per example: the ten launches of a step are independent, so they run concurrently and
    # the chip is not left idle while one example's longest strings finish
    main = torch.cuda.current_stream(device)
    ev_fork = torch.cuda.Event(enable_timing=True)
    ev_join = torch.cuda.Event(enable_timing=True)
    ev_done = {ex: torch.cuda.Event() for ex in shards}
    res_pos, pos = {}, 0
    for ex, sh in shards.items():
        res_pos[ex] = pos
        pos += sh["n"]
    # batching policy (setup, untimed): one back-to-back pass measures each example's kernel time; the launches of a
    # step are then spread over --streams HIP streams longest-first, each onto the least loaded stream, so that the
    # long examples start at once and no stream is left with a long kernel at the end of the step
    order = list(shards)
    streams = {ex: main for ex in shards}

    def launch_all(order_, streams_):
        ev_fork.record(main)
        for ex in order_:
            sh, st = shards[ex], streams_[ex]
            if st is not main:
                st.wait_event(ev_fork)
            sh["img"].match_tensors(sh["bytes"], sh["off"], results[res_pos[ex]:res_pos[ex] + sh["n"]], stream=st)
            if st is not main:
                ev_done[ex].record(st)
                main.wait_event(ev_done[ex])
        ev_join.record(main)

    n_streams = 1
    if args.concurrent and args.streams != 1:
        launch_all(order, streams)                       # back to back on one stream: each example's own kernel time
        torch.cuda.synchronize()
        cost = {ex: shards[ex]["img"].last_kernel_ms(local) for ex in shards}
        pool = [torch.cuda.Stream(device) for _ in range(min(max(args.streams, 6), len(shards)))]

        def schedule(n):
            load, queue, where = [0.0] * n, [[] for _ in range(n)], {}
            for ex in sorted(shards, key=lambda e: -cost[e]):
                k = load.index(min(load))
                load[k] += cost[ex]
                queue[k].append(ex)
                where[ex] = pool[k]
            return [q[j] for j in range(max(len(q) for q in queue)) for q in queue if j < len(q)], where

        # --streams 0: the stream count is tried out (how well kernels overlap depends on which ones meet): two
        # untimed passes per candidate, the fastest fork-to-join span wins
        best = None
        for n in ([args.streams] if args.streams > 1 else [3, 4, 6]):
            cand = schedule(min(n, len(shards)))
            spans = []
            for _ in range(2 if args.streams < 1 else 0):
                launch_all(*cand)
                ev_join.synchronize()
                spans.append(ev_fork.elapsed_time(ev_join))
            t = min(spans) if spans else 0.0
            if best is None or t < best[0]:
                best = (t, n, cand)
        n_streams, (order, streams) = best[1], best[2]

    def step(record):
        launch_all(order, streams)
        bitmap = sharding.pack_bitmap(results)
        if dist:
            dist.gather(bitmap.to(comm_dev), gathered, dst=0)      # RCCL over xGMI: the path's only exchange
        if record:
            for ex, sh in shards.items():
                kernel_ms[ex].append(sh["img"].last_kernel_ms(local))
            ev_join.synchronize()
            span_ms.append(ev_fork.elapsed_time(ev_join))
        return bitmap

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
        step(True)
    fence()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_bytes * world * args.steps / dt / 1e9
        # roofline of the dominant kernel (mfa_walk_kernel): algorithmic bytes of one step's launches
        # (1 B per input character + 8 B offset + 1 B result per string) over their summed durations
        alg = total_bytes + 9 * total_strings
        # the ten launches overlap: the device time of a step's match work is the span from the fork event
        # (recorded before the first launch) to the join event (after the last kernel), both HIP events on
        # the streams the kernels run on
        kern_s = float(np.mean(span_ms)) * 1e-3
        achieved = alg / kern_s / 1e9
        kinds = {capi.KERNEL_GENERIC: "mfa_walk_kernel", capi.KERNEL_SPECIALISED: "mfa_jit_kernel"}
        per_ex = {str(ex): {"kernel_ms": float(np.mean(kernel_ms[ex])), "bytes": shards[ex]["nbytes"],
                            "GB/s": shards[ex]["nbytes"] / (float(np.mean(kernel_ms[ex])) * 1e-3) / 1e9,
                            "accepted": None} for ex in shards}
        pos = 0
        for ex, sh in shards.items():
            per_ex[str(ex)]["accepted"] = int(results[pos:pos + sh["n"]].sum().item())
            pos += sh["n"]
        # HBM traffic of one step from the PMC passes kept under profiles/ (collected separately: counters cannot be
        # read from inside the run); only quoted when this run is the workload they were collected on
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01k_traffic.json")) as f:
                tj = json.load(f)
            wl = tj["workload"]
            if (wl["strings_per_example"], wl["min_len"], wl["max_len"]) == (n_per, args.min_len, args.max_len) and args.concurrent:
                traffic = tj["hbm_bytes_per_step"]
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "input GB/s (chars matched/sec) on 10-example attack corpus",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "10 README MFA examples (plain mode), %d pumped attack strings per example per GPU, "
                                   "pump size log-uniform [%d, %d], alternating with/without suffix "
                                   "(BASELINE configs[3] shard: 10M strings over 8 GPUs)" % (n_per, args.min_len, args.max_len),
                       "strings_per_example": n_per, "min_len": args.min_len, "max_len": args.max_len, "strings_per_gpu": total_strings, "bytes_per_gpu": total_bytes, "parallelism": "dp%d" % world,
                       "exchange": "gather of the result bitmap to rank 0" + (" (RCCL)" if dist else " (single rank: none)")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "%s: 10 launches per step, one per example, %s" % (
                             "/".join(sorted({kinds[sh["img"].info()["last_kernel"]] for sh in shards.values()})),
                             "concurrent on %d streams, longest first (duration = fork-to-join span)" % n_streams if args.concurrent else "back to back on one stream"),
                         "algorithmic_bytes_per_step": alg,
                         # mean duration of one of the step's dispatches (own start/stop events; they overlap): what a kernel trace's
                         # per-dispatch durations of the timed steps average to (profiles/*_span_check.txt lists their sum per step)
                         "dispatch_ms_mean": float(np.mean([np.mean(kernel_ms[ex]) for ex in shards])), "kernel_seconds_per_step": kern_s},
            "per_example": per_ex,
        }
        if not args.no_cpu_baseline and world == 1:
            pos, gpu_res = 0, {}
            for ex, sh in shards.items():
                gpu_res[ex] = results[pos:pos + PARITY_N].cpu().numpy()
                pos += sh["n"]
            out["cpu_baseline"] = cpu_baseline(shards, gpu_res)
        if not args.no_secondary and world == 1:
            for sh in shards.values():                   # free the headline shard before the other configurations
                sh["bytes"] = sh["off"] = None
            torch.cuda.empty_cache()
            out["secondary"] = [secondary_dfa(device), secondary_config3(device)] + secondary_config5(device)
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
