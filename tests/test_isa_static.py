"""Static checks on the gfx950 assembly of the hand-scheduled kernels (hipcc cross-compiles without a GPU).

The scalar-base LDS-DMA copy (conv_lds_common.h glds16_s) takes its 64-bit base in an SGPR pair through an inline-asm "s"
constraint.  If the compiler decides the value lives in VGPRs -- it did once: integer divisions are expanded into vector
code and everything derived from their results followed them there -- the statement assembles into
`global_load_lds_dwordx4 v1, v[4:5]`, which the assembler rejects only in some builds and which would read per-lane
garbage bases in the others.  So: every copy of the third-generation kernels must name an SGPR pair."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", ["conv_lds2.hip", "conv_wgrad2.hip"])
def test_scalar_base_copies_take_sgpr_bases(src, tmp_path):
    out = tmp_path / "k.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-S", "--cuda-device-only", "-o", str(out),
                    os.path.join(ROOT, "cotnet_amd", "csrc", src)], check=True, capture_output=True)
    copies = [ln.strip() for ln in open(out) if "global_load_lds_dwordx4" in ln]
    assert len(copies) > 50, "the kernels of this file issue LDS-DMA copies"
    bad = [c for c in copies if not re.search(r"global_load_lds_dwordx4 v\d+, s\[\d+:\d+\]$", c)]
    assert not bad, bad[:5]
    shutil.rmtree(tmp_path, ignore_errors=True)
