"""Same-box comparison of mmf_gemm_bf16 with the vendor library GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the twelve
GEMM shapes of one VisualBERT VQA2 encoder layer (forward / dgrad / wgrad).  The library kernels carry no epilogue, so
this is a ceiling for the K-loop at these shapes, not a like-for-like step time.  Prints one line per shape and a JSON
summary (written to gpurun_out/gemm_vs_library.json when that directory exists)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmf_amd import _native as nat
from tools.gemm_bench import SHAPES

DEV = "cuda"
ROT = 6      # operand sets rotated through, so consecutive launches do not find their operands in L2


def timed(fs, iters=24):
    for f in fs:
        f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters):
        fs[i % len(fs)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def make(kind, m, n, k):
    mine, lib = [], []
    for _ in range(ROT):
        if kind == "NT":
            A = torch.randn(m, k, device=DEV).bfloat16(); B = torch.randn(n, k, device=DEV).bfloat16()
            C = torch.empty(m, n, device=DEV, dtype=torch.bfloat16)
            mine.append(lambda A=A, B=B, C=C: nat.gemm(A, B, C, m, n, k, k, k, n))
            lib.append(lambda A=A, B=B, C=C: torch.matmul(A, B.t(), out=C))
        elif kind == "NN":
            A = torch.randn(m, k, device=DEV).bfloat16(); B = torch.randn(k, n, device=DEV).bfloat16()
            C = torch.empty(m, n, device=DEV, dtype=torch.bfloat16)
            mine.append(lambda A=A, B=B, C=C: nat.gemm(A, B, C, m, n, k, k, n, n, b_kmajor=True))
            lib.append(lambda A=A, B=B, C=C: torch.matmul(A, B, out=C))
        else:
            A = torch.randn(k, m, device=DEV).bfloat16(); B = torch.randn(k, n, device=DEV).bfloat16()
            C = torch.empty(m, n, device=DEV, dtype=torch.float32); C16 = torch.empty(m, n, device=DEV, dtype=torch.bfloat16)
            mine.append(lambda A=A, B=B, C=C: nat.gemm(A, B, C, m, n, k, m, n, n, a_kmajor=True, b_kmajor=True))
            lib.append(lambda A=A, B=B, C=C16: torch.matmul(A.t(), B, out=C))
    return mine, lib


if __name__ == "__main__":
    rows = []
    tot = {"mine": 0.0, "lib": 0.0}
    flops = 0.0
    try:
        pref = str(torch.backends.cuda.preferred_blas_library())
    except Exception as e:      # noqa: BLE001
        pref = "unknown (%s)" % e
    for name, kind, m, n, k in SHAPES:
        mine, lib = make(kind, m, n, k)
        t_mine, t_lib = timed(mine), timed(lib)
        fl = 2.0 * m * n * k
        flops += fl; tot["mine"] += t_mine; tot["lib"] += t_lib
        rows.append(dict(name=name, kind=kind, M=m, N=n, K=k, mine_us=round(t_mine * 1e3, 1), lib_us=round(t_lib * 1e3, 1),
                         mine_tflops=round(fl / t_mine / 1e9, 1), lib_tflops=round(fl / t_lib / 1e9, 1)))
        print("%-11s %s M=%5d N=%5d K=%5d  mine %7.1f us %6.1f TF | library %7.1f us %6.1f TF" % (
            name, kind, m, n, k, t_mine * 1e3, fl / t_mine / 1e9, t_lib * 1e3, fl / t_lib / 1e9), flush=True)
        del mine, lib
        torch.cuda.empty_cache()
    summary = dict(blas=pref, rotate=ROT, layer_us=dict(mine=round(tot["mine"] * 1e3, 1), library=round(tot["lib"] * 1e3, 1)),
                   layer_tflops=dict(mine=round(flops / tot["mine"] / 1e9, 1), library=round(flops / tot["lib"] / 1e9, 1)), shapes=rows)
    print(json.dumps(summary))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        json.dump(summary, open(os.path.join(out, "gemm_vs_library.json"), "w"), indent=1)
