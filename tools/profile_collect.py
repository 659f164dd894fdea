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

n_region = bench["roofline"]["region_scan_kernel"]["launches_per_step"]
WALK = ("walk_kernel", "mfa_jit_kernel")
trace_rows = list(csv.DictReader(open(one("stats/**/*kernel_trace.csv"))))
# walk launches per step: the headline steps come first in the trace (warm-up + timed steps, no set-up pass with the table engine)
n_steps_traced = bench["warmup"] + bench["steps"]
walk_rows = sorted((r for r in trace_rows if any(w in r["Kernel_Name"] for w in WALK)), key=lambda r: int(r["Start_Timestamp"]))
region_rows = sorted((r for r in trace_rows if "region_scan_kernel" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
t_end_steps = int(region_rows[n_region * n_steps_traced - 1]["End_Timestamp"]) if len(region_rows) >= n_region * n_steps_traced else None
n_walk = max(1, round(sum(1 for r in walk_rows if t_end_steps is None or int(r["Start_Timestamp"]) <= t_end_steps + 2000000) / n_steps_traced))


def pmc_sum(counter):
    """the counter summed over the dispatches of the run's LAST step: its region launches and its walk launches (the run's
    earlier dispatches are the untimed set-up pass)"""
    rows = [r for r in csv.DictReader(open(one(counter + "/**/*counter_collection.csv"))) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    walk = [r for r in rows if any(w in r["Kernel_Name"] for w in WALK)]
    region = [r for r in rows if "region_scan_kernel" in r["Kernel_Name"]]
    assert len(walk) >= 1 and len(region) >= n_region, (counter, len(walk), len(region))
    # the PMC runs execute exactly one step (--steps 1 --warmup 0, no secondary lines): every dispatch of these kernels is that step's
    return (sum(float(r["Counter_Value"]) for r in region), sum(float(r["Counter_Value"]) for r in walk), len(region) + len(walk))


fetch_r, fetch_w, n_f = pmc_sum("FETCH_SIZE")
write_r, write_w, n_w = pmc_sum("WRITE_SIZE")
fetch, write = fetch_r + fetch_w, write_r + write_w
alg = bench["roofline"]["algorithmic_bytes_per_step"]
cfg = bench["config"]
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd"))
from mfa_amd import srchash
traffic = {
    "round": int(tag[1:3]) if tag[1:3].isdigit() else 0, "tag": tag,
    "csrc_sha16": srchash.kernel_source_hash(),      # bench.py quotes these counters only while the kernel sources still hash to this
    "workload": {"strings_per_example": cfg.get("strings_per_example", 125000), "min_len": cfg.get("min_len", 1024),
                 "max_len": cfg.get("max_len", 65536), "n_gpus": bench["n_gpus"]},
    "how": "two separate rocprofv3 --pmc passes over `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary` (FETCH_SIZE, then "
           "WRITE_SIZE), summed over the %d region_scan_kernel and %d walk dispatches of the run's one step; units are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the "
           "bytes of wide coalesced reads)" % (n_region, n_walk),
    "fetch_size_kib": fetch, "write_size_kib": write,
    "region_scan_kernel": {"fetch_size_kib": fetch_r, "write_size_kib": write_r, "hbm_bytes": int(2 * fetch_r * 1024 + write_r * 1024)},
    "walk_kernels": {"fetch_size_kib": fetch_w, "write_size_kib": write_w, "hbm_bytes": int(2 * fetch_w * 1024 + write_w * 1024)},
    "hbm_bytes_per_step": int(2 * fetch * 1024 + write * 1024),
    "algorithmic_bytes_per_step": alg,
}
with open(os.path.join(dst, tag + "_traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
    f.write("\n")
span = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "span_from_trace.py"), one("stats/**/*kernel_trace.csv"), str(n_region), str(n_walk), "0", str(bench["warmup"]), str(bench["steps"])],
                      capture_output=True, text=True).stdout
with open(os.path.join(dst, tag + "_span_check.txt"), "w") as f:
    f.write(span)
print(span)
print(json.dumps(traffic, indent=1))
print("bench:", bench["value"], bench["unit"], bench["ms_per_step"], "ms/step; roofline", bench["roofline"])
