"""Build-time invariants of regions.hip that its hand-placed `s_waitcnt vmcnt(n)` rely on (see the comment above `struct Row`):
no register spills (a spill is a vector memory operation the counted waits do not know about) and eight waves per SIMD."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_region_kernels_have_no_spills(tmp_path):
    src = os.path.join(ROOT, "re2-modification_amd", "csrc", "regions.hip")
    p = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-x", "hip", "-S",
                        "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "regions.s"), src],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", p.stderr)[1:]
    kernels = [b for b in blocks if "region_scan_kernel" in b.split()[0]]
    assert len(kernels) >= 4                                   # streaming-only + rotation depths 2, 3, 4
    for b in kernels:
        name = b.split()[0]
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        vgprs = int(re.search(r" VGPRs: (\d+)", b).group(1))
        occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
        assert scratch == 0 and vgprs <= 64 and occ == 8, (name, scratch, vgprs, occ)
    asm = (tmp_path / "regions.s").read_text()
    assert "scratch_" not in asm and "buffer_store" not in asm
    # every request is two loads, every counted wait leaves a multiple of two outstanding
    assert len(re.findall(r"s_waitcnt vmcnt\((2|4|6)\)", asm)) >= 6
