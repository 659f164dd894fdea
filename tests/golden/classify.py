#!/usr/bin/env python3
"""Fixture sensitivity classes of every golden (automaton, string) pair -> tests/golden/classes.json.

  C  the reference on glibc's heap (REF_HARNESS_ALLOC=glibc: no arena, strings matched one after the other in one process, as
     `./diploma -match` does) answers differently from the canonical bump-arena run the goldens hold.  Recorded with both
     answers; excluded from pass/fail nowhere in this repository (the GPU path implements the canonical model), listed so that
     a user who compares with a stock build of the reference knows where to expect a difference.
  B  not C, and tie-sensitive: the CPU restatement with "the LAST state that ties on (pos, node) wins" instead of the
     reference's "first wins" (mfa.cpp:206-211) answers differently.
  A  everything else (the answer survives flipping the tie-break).
Memory-less automata have no ties: all A.  Needs oracle/_ref/ref_harness (build container only).
Usage: python tests/golden/classify.py   (from the repo root)"""
import ctypes
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "re2-modification_amd"))
import numpy as np  # noqa: E402
import oracle_lib  # noqa: E402
from mfa_amd import image  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


def main():
    with open(os.path.join(HERE, "manifest.json")) as f:
        manifest = json.load(f)
    lib = oracle_lib.lib()
    lib.mfa_oracle_set_tie_policy.argtypes = [ctypes.c_int]
    out = {"how": __doc__.split("Usage")[0].strip(), "automata": {}}
    tot = {"A": 0, "B": 0, "C": 0}
    for auto in manifest["automata"]:
        blob = image.blob_from_dump(oracle_lib.load_dump(auto["name"]))
        is_mfa = image.blob_info(blob)["kind"] == image.KIND_MFA
        entry = {"counts": {"A": 0, "B": 0, "C": 0}, "B": {}, "C": {}}
        for sset in auto["sets"]:
            strings = oracle_lib.load_set(sset)
            want = oracle_lib.load_bits(auto["name"], sset)
            text = b"".join(s + b"\n" for s in strings)
            env = dict(os.environ, REF_HARNESS_ALLOC="glibc")
            p = subprocess.run([HARNESS, "match", auto["mode"], auto["regex"]], input=text, capture_output=True, env=env, timeout=1200)
            assert p.returncode == 0, (auto["name"], sset, p.stderr[:200])
            glibc = np.array([int(c) for c in p.stdout.split()], dtype=np.uint8)
            assert glibc.size == want.size
            c_idx = np.nonzero(glibc != want)[0]
            b_idx = np.array([], dtype=np.int64)
            if is_mfa:
                lib.mfa_oracle_set_tie_policy(1)
                try:
                    last = oracle_lib.OracleImage(blob).match(strings)
                finally:
                    lib.mfa_oracle_set_tie_policy(0)
                b_idx = np.setdiff1d(np.nonzero(last != want)[0], c_idx)
            if c_idx.size:
                entry["C"][sset] = [{"index": int(k), "string": strings[k].decode("latin1"), "bump": int(want[k]), "glibc": int(glibc[k])} for k in c_idx]
            if b_idx.size:
                entry["B"][sset] = [int(k) for k in b_idx]
            entry["counts"]["C"] += int(c_idx.size)
            entry["counts"]["B"] += int(b_idx.size)
            entry["counts"]["A"] += int(want.size - c_idx.size - b_idx.size)
        for k in tot:
            tot[k] += entry["counts"][k]
        out["automata"][auto["name"]] = entry
        print(auto["name"], entry["counts"], flush=True)
    out["totals"] = tot
    with open(os.path.join(HERE, "classes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("totals", tot)


if __name__ == "__main__":
    main()
