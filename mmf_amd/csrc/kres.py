"""Print per-kernel register / scratch / LDS usage (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
from mmf_amd.csrc.build import FLAGS, HIPCC
for src in sys.argv[1:] or ["gemm.hip", "attention.hip", "rowops.hip"]:
    r = subprocess.run([HIPCC, *FLAGS, "-c", os.path.join(HERE, src), "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^ ]+ +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs|VGPRs Spill): (.*?) \[-R", line)
        m = m or re.search(r": +(Function Name|Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (.*?) \[-R", line)
        if not m: continue
        k, v = m.group(1), m.group(2)
        if k in ("Function Name", "Name"):
            if cur: print(cur)
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            cur = {"kernel": re.sub(r"\(anonymous namespace\)::", "", name)[:90]}
        else:
            cur[k.split(" [")[0]] = v
    if cur: print(cur)
