"""Micro-benchmark of mmf_gemm_bf16 on the VisualBERT VQA2 shapes (forward / dgrad / wgrad)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmf_amd import _native as nat

M = 7296
SHAPES = [  # (name, kind, M, N, K)
    ("qkv fwd", "NT", M, 2304, 768), ("out fwd", "NT", M, 768, 768), ("ffn1 fwd", "NT", M, 3072, 768), ("ffn2 fwd", "NT", M, 768, 3072),
    ("qkv dgrad", "NN", M, 768, 2304), ("out dgrad", "NN", M, 768, 768), ("ffn1 dgrad", "NN", M, 768, 3072), ("ffn2 dgrad", "NN", M, 3072, 768),
    ("qkv wgrad", "TN", 2304, 768, M), ("out wgrad", "TN", 768, 768, M), ("ffn1 wgrad", "TN", 3072, 768, M), ("ffn2 wgrad", "TN", 768, 3072, M),
]


def run(kind, m, n, k, iters=20):
    dev = "cuda"
    if kind == "NT":
        A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16()
        C = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        f = lambda: nat.gemm(A, B, C, m, n, k, k, k, n)
    elif kind == "NN":
        A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16()
        C = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        f = lambda: nat.gemm(A, B, C, m, n, k, k, n, n, b_kmajor=True)
    else:
        A = torch.randn(k, m, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16()
        C = torch.empty(m, n, device=dev, dtype=torch.float32)
        f = lambda: nat.gemm(A, B, C, m, n, k, m, n, n, a_kmajor=True, b_kmajor=True)
    for _ in range(3):
        f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * m * n * k / ms / 1e9


if __name__ == "__main__":
    tot_ms = tot_fl = 0
    only = sys.argv[1:]
    for name, kind, m, n, k in SHAPES:
        if only and not any(o in name for o in only):
            continue
        ms, tf = run(kind, m, n, k)
        tot_ms += ms; tot_fl += 2.0 * m * n * k
        print("%-12s %s M=%5d N=%5d K=%5d  %8.1f us  %7.1f TFLOP/s" % (name, kind, m, n, k, ms * 1e3, tf))
    print("layer total %.1f us, %.1f TFLOP/s" % (tot_ms * 1e3, tot_fl / tot_ms / 1e9))
