#!/usr/bin/env python3
"""Development: the interesting fields of a bench.py JSON line.  usage: bench_summary.py file.json"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
r = d["roofline"]
print("value %.0f GB/s  ms/step %.3f  frac %.3f (achieved %.0f)  region: %.3f ms/step frac %.3f  traffic %s" % (
    d["value"], d["ms_per_step"], r["frac"], r["achieved"], r["region_scan_kernel"]["ms_per_step"], r["region_scan_kernel"]["frac"], r.get("traffic")))
print("kernel:", r["kernel"])
if d.get("parity_sample"):
    print("parity sample:", {k: d["parity_sample"][k] for k in ("strings", "bytes", "max_len", "mismatches", "cores", "seconds")})
c = d.get("cpu_baseline")
if c:
    print("cpu_baseline: %.6f GB/s kind %s cores %d parity %s | %s" % (c["value"], c["kind"], c["cores"], c["parity"], c["sample"]))
    if c.get("restatement_all_cores"):
        print("  restatement all cores: %.4f GB/s on %d cores" % (c["restatement_all_cores"]["value"], c["restatement_all_cores"]["cores"]))
for s in d.get("secondary", []):
    keys = [k for k in ("GB/s", "GB/s_on_sum_of_lengths", "char_steps_per_s", "region_ms", "walk_ms", "span_ms", "kernel_ms") if k in s]
    par = s.get("parity_sample") or s.get("parity_oracle_sample") or {}
    print("- %s\n    %s | kernel %s | parity %s" % (s["workload"], "  ".join("%s=%.4g" % (k, s[k]) for k in keys), s.get("kernel"),
                                                 {k: par[k] for k in ("strings", "max_len", "mismatches") if k in par}))
    if "frac_of_hbm_peak_on_touched_bytes" in s:
        print("    frac on touched bytes:", s["frac_of_hbm_peak_on_touched_bytes"])
    if "mixed_call" in s:
        print("    mixed call:", s["mixed_call"])
