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
