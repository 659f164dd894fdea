"""GPU parity: the HIP path (through the C-ABI) against the golden vectors produced by the
reference and against the CPU restatement, bit-exact 0/1."""
import json
import os

import numpy as np
import pytest

import oracle_lib
from mfa_amd import capi, image

pytestmark = pytest.mark.gpu

with open(os.path.join(oracle_lib.GOLDEN, "manifest.json")) as f:
    MANIFEST = json.load(f)


def gpu_match(img, strings):
    import torch
    data, off = oracle_lib.pack(strings)
    d_bytes = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
    d_bytes[:len(data)] = torch.from_numpy(data)
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    res = img.match_tensors(d_bytes, d_off)
    torch.cuda.synchronize()
    return res.cpu().numpy()


@pytest.mark.parametrize("auto", MANIFEST["automata"], ids=lambda a: a["name"])
def test_golden(auto):
    blob = image.blob_from_dump(oracle_lib.load_dump(auto["name"]))
    img = capi.Image(blob)
    for sset in auto["sets"]:
        strings = oracle_lib.load_set(sset)
        want = oracle_lib.load_bits(auto["name"], sset)
        got = gpu_match(img, strings)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s/%s: %d mismatches, first %r want %d got %d" % (
            auto["name"], sset, bad.size, strings[bad[0]], want[bad[0]], got[bad[0]])
    img.close()


def test_host_entry_point():
    img = capi.Image(image.blob_from_dump(oracle_lib.load_dump("ex1_plain")))
    data, off = oracle_lib.pack([b"aa", b"aaa", b"aaaa", b"b", b"aaaaaab", b"ab", b"aaaaaaaa", b""])
    assert list(img.match_host(data, off)) == [1, 1, 1, 0, 0, 0, 1, 1]
