"""Kernel resource report: VGPR / AGPR / SGPR / LDS / scratch of every kernel in mmf_amd/csrc/*.o (read from the code object's
metadata notes).  python tools/kres.py [substring]   — exits non-zero if any kernel spills to scratch."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = "/opt/rocm/lib/llvm/bin"


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    tmp = "/tmp/kres_co"
    os.makedirs(tmp, exist_ok=True)
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    notes = ""
    for src in sorted(os.listdir(os.path.join(ROOT, "mmf_amd", "csrc"))):
        if not src.endswith(".o"):
            continue
        fat, out = os.path.join(tmp, src + ".fat"), os.path.join(tmp, src + ".co")
        subprocess.run([os.path.join(TOOLS, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat,
                        os.path.join(ROOT, "mmf_amd", "csrc", src), os.path.join(tmp, "x.o")], capture_output=True)
        if not os.path.exists(fat):
            continue
        subprocess.run([os.path.join(TOOLS, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], capture_output=True, text=True)
        notes += subprocess.run([os.path.join(TOOLS, "llvm-readelf"), "--notes", out], capture_output=True, text=True).stdout
    bad = 0
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        if pat and pat not in name:
            continue
        agpr = blk.strip().split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        spill = g("vgpr_spill_count"); scratch = g("private_segment_fixed_size")
        flag = "" if (spill in ("0", "?") and scratch in ("0", "?")) else "   <-- SPILL"
        bad += bool(flag)
        print("vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s  %s%s" % (g("vgpr_count"), agpr, g("sgpr_count"),
              g("group_segment_fixed_size"), scratch, dem[:150], flag))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
