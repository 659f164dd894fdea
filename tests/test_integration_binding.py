"""INTEGRATION.md, binding B: integration/gpu_match.cpp compiled against the reference's own headers and objects (build
container only: needs /root/reference and oracle/_ref/*.o), and the blob its freeze() makes from the reference's graphs
compared with the committed golden images."""
import json
import os
import subprocess

import pytest

import oracle_lib
from mfa_amd import image

REF = "/root/reference"
OBJ = os.path.join(oracle_lib.ROOT, "oracle", "_ref")
OBJS = ["automata.o", "mfa.o", "bt_binary_tree.o", "bt_bt_thomson.o", "bt_bt_glushkov.o", "bt_bt_mfa.o", "bt_bt_ssnf.o",
        "regex_parser.o", "regex_regex.o", "regex_bnf.o", "regex_reverse.o", "regex_helpers.o"]

pytestmark = pytest.mark.skipif(not (os.path.isdir(REF) and all(os.path.exists(os.path.join(OBJ, o)) for o in OBJS)),
                                reason="needs the reference sources and their objects (build container only)")

with open(os.path.join(oracle_lib.GOLDEN, "manifest.json")) as f:
    MANIFEST = json.load(f)


@pytest.fixture(scope="module")
def freeze_check(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("binding") / "freeze_check")
    inc = os.path.join(oracle_lib.ROOT, "include")
    src = os.path.join(oracle_lib.ROOT, "integration")
    cmd = ["g++", "-std=c++17", "-w", "-I" + REF, "-I" + inc, "-o", out, os.path.join(src, "freeze_check.cpp"), os.path.join(src, "gpu_match.cpp")]
    cmd += [os.path.join(OBJ, o) for o in OBJS]
    cmd += ["-L" + os.path.join(oracle_lib.ROOT, "re2-modification_amd", "csrc"), "-lmfa_hip",
            "-Wl,-rpath," + os.path.join(oracle_lib.ROOT, "re2-modification_amd", "csrc")]
    subprocess.check_call(cmd)
    return out


@pytest.mark.parametrize("auto", [a for a in MANIFEST["automata"] if a["mode"] in ("plain", "bnf", "reverse", "thompson", "glushkov")],
                         ids=lambda a: a["name"])
def test_freeze_equals_golden_image(auto, freeze_check, tmp_path):
    p = subprocess.run([freeze_check, auto["mode"], auto["regex"]], capture_output=True, text=True, cwd=tmp_path, timeout=60)
    assert p.returncode == 0, p.stderr
    want = image.blob_from_dump(oracle_lib.load_dump(auto["name"]))
    assert bytes.fromhex(p.stdout.strip()) == bytes(want)
