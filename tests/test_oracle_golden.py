"""The CPU restatement (oracle/mfa_oracle.c) against the reference's own answers.

tests/golden/ was produced by running the reference itself (oracle/_ref/ref_harness,
canonical allocation-order mode) -- see tests/golden/make_golden.py.  This is what
pins the oracle; the GPU parity tests then compare the HIP path with the oracle.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib
from mfa_amd import image

with open(os.path.join(oracle_lib.GOLDEN, "manifest.json")) as f:
    MANIFEST = json.load(f)


@pytest.mark.parametrize("auto", MANIFEST["automata"], ids=lambda a: a["name"])
def test_restatement_matches_reference(auto):
    blob = image.blob_from_dump(oracle_lib.load_dump(auto["name"]))
    img = oracle_lib.OracleImage(blob)
    for sset in auto["sets"]:
        strings = oracle_lib.load_set(sset)
        want = oracle_lib.load_bits(auto["name"], sset)
        assert len(want) == len(strings) == MANIFEST["sets"][sset]
        got = img.match(strings)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s/%s: %d mismatches, first %r want %d" % (
            auto["name"], sset, bad.size, strings[bad[0]], want[bad[0]])


def test_survey_anchors():
    """SURVEY.md section 8c sanity anchors (measured on the unmodified reference)."""
    ex1 = oracle_lib.OracleImage(image.blob_from_dump(oracle_lib.load_dump("ex1_plain")))
    assert list(ex1.match([b"aa", b"aaa", b"aaaa", b"b", b"aaaaaab", b"ab", b"aaaaaaaa"])) == [1, 1, 1, 0, 0, 0, 1]
    abb = oracle_lib.OracleImage(image.blob_from_dump(oracle_lib.load_dump("nfa_abb_plain")))
    assert list(abb.match([b"abb", b"aabb", b"ab", b"bbbbabb"])) == [1, 1, 0, 1]


def test_fixture_classes():
    """tests/golden/classes.json (tests/golden/classify.py): every golden pair is tagged A (tie-insensitive), B (tie-sensitive,
    glibc run == canonical run) or C (the reference on glibc's heap answers differently; both answers recorded).  The B tags are
    re-derived here: flipping the tie-break of the CPU restatement flips exactly the B and C strings' neighbourhood."""
    import ctypes
    with open(os.path.join(oracle_lib.GOLDEN, "classes.json")) as f:
        classes = json.load(f)
    with open(os.path.join(oracle_lib.GOLDEN, "manifest.json")) as f:
        manifest = json.load(f)
    lib = oracle_lib.lib()
    lib.mfa_oracle_set_tie_policy.argtypes = [ctypes.c_int]
    total = 0
    for auto in manifest["automata"]:
        entry = classes["automata"][auto["name"]]
        n = sum(len(oracle_lib.load_set(s)) for s in auto["sets"])
        assert sum(entry["counts"].values()) == n
        total += n
        if not entry["B"] and not entry["C"]:
            continue
        blob = image.blob_from_dump(oracle_lib.load_dump(auto["name"]))
        for sset in auto["sets"]:
            strings = oracle_lib.load_set(sset)
            want = oracle_lib.load_bits(auto["name"], sset)
            lib.mfa_oracle_set_tie_policy(1)
            try:
                last = oracle_lib.OracleImage(blob).match(strings)
            finally:
                lib.mfa_oracle_set_tie_policy(0)
            flipped = set(int(k) for k in np.nonzero(last != want)[0])
            c_here = set(c["index"] for c in entry["C"].get(sset, []))
            assert flipped - c_here == set(entry["B"].get(sset, [])), (auto["name"], sset)
            for c in entry["C"].get(sset, []):
                assert c["bump"] == int(want[c["index"]]) and c["glibc"] != c["bump"]
    assert sum(classes["totals"].values()) == total and classes["totals"]["C"] < 100
