#!/usr/bin/env python3
"""Development (no GPU): executed steps of the walk's host emulation (tests/emul) per automaton on pumped strings -- what a change to the
probe control does to the step counts.  usage: emul_steps.py [emulator binary] [names ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
from mfa_amd import image, corpus
exe = sys.argv[1] if len(sys.argv) > 1 else "/tmp/w/walk_emul"
names = sys.argv[2:] or ["ex%d_plain" % k for k in range(1, 11)] + ["ex3_reverse", "ex6_reverse", "ex8_reverse", "ex8_bnf", "ex2_reverse", "ex9_reverse", "ex10_reverse"]
tot = 0
for name in names:
    ex = int("".join(c for c in name.split("_")[0] if c.isdigit()))
    regex, pump, suffix, prefix = corpus.ALL_EXAMPLES[ex]
    blob = "/tmp/w/%s.blob" % name
    open(blob, "wb").write(image.blob_from_dump(oracle_lib.load_dump(name)))
    strings = []
    for n in (1100, 3000, 9000, 30000, 65536):
        for ws in (False, True):
            strings.append(prefix + corpus.pumped_string(n, pump) + (suffix if ws else ""))
    p = subprocess.run([exe, blob, "8", "1"], input=("\n".join(strings) + "\n").encode(), capture_output=True)
    line = [l for l in p.stderr.decode().splitlines() if l.startswith("emul:") and "steps" in l][0]
    steps = int(line.split("steps ")[1].split(",")[0])
    tot += steps
    print("%-14s %s  answers %s" % (name, line[6:], p.stdout.decode().replace("\n", "")))
print("total steps", tot)
