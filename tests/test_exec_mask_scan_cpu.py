"""The hipcc hazard of profiles/HISTORY.md's wide-net section, as a build-time check: no register save (v_accvgpr_write / scratch_store) of an outer
value under a reduced EXEC mask in the (256, h2) minibatch kernels' assembly (tools/exec_mask_scan.py).  Compiles two of the eight
translation units to assembly (hipcc cross-compiles without a GPU; ~20 s each)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "elegantrl_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("unit", ["ppo_step_wd_12.hip", "ppo_step_wd3_12.hip"])
def test_no_register_save_under_a_reduced_exec_mask(unit, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exec_mask_scan
    out = tmp_path / (unit + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form",
                           "-DERL_SLAB_ST=1", "--cuda-device-only", "-S", "-o", str(out), unit], cwd=CSRC, stderr=subprocess.DEVNULL)
    bad, lines = exec_mask_scan.scan(str(out))
    assert bad == 0, "\n".join(lines[:10])
    asm = out.read_text()
    assert "scratch_load" not in asm and "scratch_store" not in asm, "the kernel spills registers to scratch memory"
