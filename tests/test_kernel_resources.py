"""Build-time guard on the hot kernel's register allocation (no GPU needed: hipcc cross-compiles): the production
instantiations of phd_update_merge_kernel must fit four waves per SIMD (<= 128 VGPRs) WITHOUT spilling to scratch —
a spill turns LDS-resident work into HBM traffic (measured once in round 1: 42 MB -> 168 MB per launch at
4096 x 256 x 64) and silently costs a few per cent."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cuda-phdslam_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
def test_production_kernels_do_not_spill():
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
           "--offload-arch=gfx950", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c",
           os.path.join(SRC, "phd_kernels.hip"), "-o", os.devnull]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=SRC)
    assert r.returncode == 0, r.stderr[-2000:]
    text = r.stderr + r.stdout
    found = {}
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)", text, re.S):
        found[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    # <STAMPS, FUSEW, CPHD>: the staged / multi-GPU step, the fused single-GPU step, the CPHD variant
    # ... and the fused CPHD step
    for tag in ("ILb0ELb0ELb0E", "ILb0ELb1ELb0E", "ILb0ELb0ELb1E", "ILb0ELb1ELb1E"):
        names = [n for n in found if "phd_update_merge_kernel" + tag in n]
        assert len(names) == 1, (tag, sorted(found))
        vgprs, scratch = found[names[0]]
        assert vgprs <= 128, (names[0], vgprs)
        assert scratch == 0, (names[0], scratch)
