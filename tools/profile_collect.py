#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/<tag>_*: kernel stats, bench line, HBM traffic, span check.
usage: profile_collect.py <tag>"""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")

def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    assert f, pattern
    return f[0]

shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
line = open(os.path.join(src, "bench.json")).read().strip()
bench = json.loads(line)
with open(os.path.join(dst, tag + "_bench.json"), "w") as f:
    f.write(line + "\n")

def pmc_sum(counter):
    rows = [r for r in csv.DictReader(open(one(counter + "/**/*counter_collection.csv")))
            if r["Kernel_Name"].startswith("mfa_jit_kernel") and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    n = bench_n_examples
    assert len(rows) >= n, (counter, len(rows))
    return sum(float(r["Counter_Value"]) for r in rows[-n:]), len(rows)       # the last step's dispatches (earlier ones: calibration pass)

bench_n_examples = len(bench["per_example"])
fetch, n_f = pmc_sum("FETCH_SIZE")
write, n_w = pmc_sum("WRITE_SIZE")
alg = bench["roofline"]["algorithmic_bytes_per_step"]
cfg = bench["config"]
traffic = {
    "round": 1, "tag": tag,
    "workload": {"strings_per_example": cfg.get("strings_per_example", 125000), "min_len": cfg.get("min_len", 1024),
                 "max_len": cfg.get("max_len", 65536), "n_gpus": bench["n_gpus"]},
    "how": "two separate rocprofv3 --pmc passes over `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary` (FETCH_SIZE, then "
           "WRITE_SIZE), summed over the %d mfa_jit_kernel dispatches of the timed step (the run's earlier %d dispatches are the untimed "
           "set-up passes: calibration and stream-count selection); units are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced "
           "reads; confirmed on dfa_tiled_kernel: 1 GiB read -> 528 695 KiB reported = 0.504x)" % (bench_n_examples, n_f - bench_n_examples),
    "fetch_size_kib": fetch, "write_size_kib": write,
    "hbm_bytes_per_step": int(2 * fetch * 1024 + write * 1024),
    "algorithmic_bytes_per_step": alg,
}
with open(os.path.join(dst, tag + "_traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
    f.write("\n")
span = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "span_from_trace.py"), one("stats/**/*kernel_trace.csv")],
                      capture_output=True, text=True).stdout
with open(os.path.join(dst, tag + "_span_check.txt"), "w") as f:
    f.write(span)
print(span)
print(json.dumps(traffic, indent=1))
print("bench:", bench["value"], bench["unit"], bench["ms_per_step"], "ms/step; roofline", bench["roofline"])
