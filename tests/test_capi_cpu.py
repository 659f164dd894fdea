"""CPU-side checks of the C-ABI library: it loads, exports every declared symbol, builds images
(host-only work) and refuses to match without a device.  No compute here."""
import json
import os
import re

import numpy as np
import pytest

import oracle_lib
from mfa_amd import capi, image

ROOT = oracle_lib.ROOT
with open(os.path.join(oracle_lib.GOLDEN, "manifest.json")) as f:
    MANIFEST = json.load(f)


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "mfa_hip.h")).read()
    declared = set(re.findall(r"\b(mfa_[a-z_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS)
    L = capi.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.mfa_version()


@pytest.mark.parametrize("auto", MANIFEST["automata"], ids=lambda a: a["name"])
def test_image_create(auto):
    blob = image.blob_from_dump(oracle_lib.load_dump(auto["name"]))
    img = capi.Image(blob)
    info = img.info()
    bi = image.blob_info(blob)
    assert info["n_nodes"] == bi["n_nodes"] and info["n_edges"] == bi["n_edges"]
    if info["kind"] == image.KIND_NFA:
        assert 2 <= info["dfa_states"] <= 4096 and info["byte_classes"] >= 1
    img.close()


def test_bad_blobs_are_refused():
    blob = bytearray(image.blob_from_dump(oracle_lib.load_dump("ex1_plain")))
    with pytest.raises(capi.MfaError) as e:
        capi.Image(bytes(blob[:20]))
    assert e.value.code == capi.ERR_BAD_BLOB
    bad = bytearray(blob); bad[0] ^= 0xFF
    with pytest.raises(capi.MfaError):
        capi.Image(bytes(bad))
    bad = bytearray(blob); bad[40 + 5 * 4 + 2] = 0x7F   # first edge's target out of range
    with pytest.raises(capi.MfaError):
        capi.Image(bytes(bad))


def test_no_cpu_fallback():
    """Without a GPU the match entry points must fail loudly, never compute on the host."""
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    img = capi.Image(image.blob_from_dump(oracle_lib.load_dump("ex1_plain")))
    data, off = oracle_lib.pack([b"aa", b"ab"])
    with pytest.raises(capi.MfaError) as e:
        img.match_host(data, off)
    assert e.value.code == capi.ERR_NO_DEVICE


def test_register_budget_of_the_built_kernels():
    """The region pass shares SIMDs with walk waves (512 VGPRs each): its streaming kernel must stay within 40 VGPRs (an allocation of 40,
    not 48) and the one-cell walk within 176 (2 x 176 + 4 x 40 = 512), or one region wave fewer fits beside two walk waves.  Read from the build's own resource
    remarks (csrc/Makefile keeps them and fails the build on the first of the two)."""
    import re
    csrc = os.path.join(os.path.dirname(capi.LIB_PATH))
    def vgprs(log, name_part):
        text = open(os.path.join(csrc, log)).read()
        out = []
        for m in re.finditer(r"Function Name: (\S+).*?\n(?:.*\n)*?.*? VGPRs: (\d+)", text):
            if name_part in m.group(1):
                out.append(int(m.group(2)))
        return out
    if not os.path.exists(os.path.join(csrc, "regions.log")):
        pytest.skip("library built without the resource remarks")
    region = vgprs("regions.log", "region_scan_kernelILi0ELi2ELb0E")
    assert region and max(region) <= 40, region
    walk1 = vgprs("walk_k1.log", "walk_kernelILi1E")
    assert walk1 and max(walk1) <= 176, walk1
