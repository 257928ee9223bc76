"""Runs the FFN-down fp32 GEMM (M = 7296, N = 768, K = 3072) through mmf_gemm_f32 and through the vendor fp32 GEMM (torch.addmm) a few times:
the command rocprofv3 --pmc wraps to compare MFMA-pipe occupancy of the two kernels (tools/pmc_run.sh; profiles/r03_fp32_gemm_pmc.txt)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmf_amd import _native as nat

M, N, K = 7296, 768, 3072
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
C = torch.empty(M, N, device="cuda")
for _ in range(10):
    nat.gemm_f32(A, W, C, M, N, K, K, K, N, bias=b)
    torch.addmm(b, A, W.t(), out=C)
A2 = torch.randn(M, 768, device="cuda"); W2 = torch.randn(3072, 768, device="cuda") * 768 ** -0.5; b2 = torch.randn(3072, device="cuda")
C2 = torch.empty(M, 3072, device="cuda")
for _ in range(10):
    nat.gemm_f32(A2, W2, C2, M, 3072, 768, 768, 768, 3072, bias=b2)
    torch.addmm(b2, A2, W2.t(), out=C2)
torch.cuda.synchronize()
