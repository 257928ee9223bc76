"""Static checks of the persistent GEMM's code objects (no GPU): its counted `s_waitcnt vmcnt(N)` waits are exact only while every vector-memory
operation of the kernel is one the source wrote - a register spill would add scratch loads / stores nobody counted."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = "/opt/rocm/lib/llvm/bin"


def _notes(obj, tmp):
    fat, co = os.path.join(tmp, "g.fat"), os.path.join(tmp, "g.co")
    subprocess.run([os.path.join(TOOLS, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(tmp, "x.o")], check=True, capture_output=True)
    subprocess.run([os.path.join(TOOLS, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
    return subprocess.run([os.path.join(TOOLS, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout


def test_persistent_kernels_use_no_scratch_and_fit_two_waves_per_simd(tmp_path):
    obj = os.path.join(ROOT, "mmf_amd", "csrc", "gemm.o")
    if not os.path.exists(obj) or not os.path.exists(os.path.join(TOOLS, "llvm-readelf")):
        pytest.skip("library objects not built here")
    notes = _notes(obj, str(tmp_path))
    seen = 0
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s*(\S+)", blk).group(1)
        if "gemm_persist_kernel" not in name:
            continue
        seen += 1
        g = lambda k: int(re.search(r"\." + k + r":\s*(\d+)", blk).group(1))
        assert g("private_segment_fixed_size") == 0, name
        assert g("vgpr_spill_count") == 0 and g("sgpr_spill_count") == 0, name
        assert g("vgpr_count") <= 256, name            # 512 threads = two waves per SIMD share its 512 registers
    assert seen == 8, seen                              # 3 tiles x 3 epilogue classes, minus 192 x 192 with a side input
