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


@pytest.mark.parametrize("jit", ["specialised", "generic"])
@pytest.mark.parametrize("auto", MANIFEST["automata"], ids=lambda a: a["name"])
def test_golden(auto, jit, monkeypatch):
    """Both MFA kernels: the automaton-specific one (where the automaton is small enough) and the
    table-driven one (MFA_JIT=0)."""
    blob = image.blob_from_dump(oracle_lib.load_dump(auto["name"]))
    is_mfa = image.blob_info(blob)["kind"] == image.KIND_MFA
    if jit == "generic":
        if not is_mfa:
            pytest.skip("memory-less automata have one kernel")
        monkeypatch.setenv("MFA_JIT", "0")
    img = capi.Image(blob)
    for sset in auto["sets"]:
        strings = oracle_lib.load_set(sset)
        want = oracle_lib.load_bits(auto["name"], sset)
        got = gpu_match(img, strings)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s/%s: %d mismatches, first %r want %d got %d" % (
            auto["name"], sset, bad.size, strings[bad[0]], want[bad[0]], got[bad[0]])
    kern = img.info()["last_kernel"]
    if not is_mfa:
        assert kern == capi.KERNEL_TABLE
    elif jit == "generic":
        assert kern == capi.KERNEL_GENERIC
    else:
        small = (image.blob_info(blob)["n_nodes"] - 1) * (1 + 3 * max(1, image.blob_info(blob)["n_cells"])) * 2 <= 224
        assert kern == (capi.KERNEL_SPECIALISED if small else capi.KERNEL_GENERIC)
    img.close()


def test_host_entry_point():
    img = capi.Image(image.blob_from_dump(oracle_lib.load_dump("ex1_plain")))
    data, off = oracle_lib.pack([b"aa", b"aaa", b"aaaa", b"b", b"aaaaaab", b"ab", b"aaaaaaaa", b""])
    assert list(img.match_host(data, off)) == [1, 1, 1, 0, 0, 0, 1, 1]


def test_cli_match_contract(tmp_path):
    """`./diploma -match`: header lines, then one 0/1 line per token, byte-identical with the reference
    (main.cpp:42-44, matchers/match.cpp:21-31)."""
    import subprocess
    diploma = os.path.join(oracle_lib.ROOT, "re2-modification_amd", "host", "diploma")
    for name in ("ex1_plain", "ex5_plain", "nfa_abb_plain", "nfa_third_plain"):
        auto = next(a for a in MANIFEST["automata"] if a["name"] == name)
        strings, want = [], []
        for sset in auto["sets"]:
            bits = oracle_lib.load_bits(name, sset)
            for s, b in zip(oracle_lib.load_set(sset), bits):
                if s:                              # the empty string cannot be a token of `cin >> text`
                    strings.append(s); want.append(b)
        text = auto["regex"].encode() + b"\n" + b"\n".join(strings) + b"\nexit\nnot-read\n"
        p = subprocess.run([diploma, "-match"], input=text, capture_output=True, cwd=tmp_path)
        assert p.returncode == 0, p.stderr
        expect = auto["header"].encode() + b"".join(b"%d\n" % b for b in want)
        assert p.stdout == expect, name
