"""Static checks on the listings of the hand-scheduled persistent kernels (no GPU needed: hipcc cross-compiles).

The inline-asm MFMA groups and the asm-issued epilogue loads rely on properties hipcc cannot check for us
(tools/check_mfma_hazard.py): no VALU write of an MFMA operand within two wait states of a nop-less MFMA, and no touch of
an asm-loaded register before its counted wait.  A scheduling change in the compiler or in the kernels shows up here."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
KERNELS = ["pg_persist_bwd.hip", "pg_persist_dec.hip", "pg_persist_dec8.hip", "pg_persist_sample.hip"]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", KERNELS)
def test_no_mfma_hazards_in_listing(src, tmp_path):
    out = tmp_path / (src + ".s")
    csrc = os.path.join(ROOT, "polygon_amd", "csrc")
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-I" + csrc,
           "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", os.path.join(csrc, src), "-o", str(out)]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_hazard.py"), str(out)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 MFMA hazards" in r.stdout and "0 early touches" in r.stdout
    listing = out.read_text()
    assert "v_mfma_f32_16x16x32_bf16" in listing          # the check looked at the real kernels
    if src == "pg_persist_bwd.hip":
        assert "PG_EPI_LOAD" in listing and "PG_EPI_WAIT" in listing
