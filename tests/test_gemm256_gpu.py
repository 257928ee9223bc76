"""The 256 x 128 three-stage-ring GEMM (experimental, debug_flags bit 14) against fp32 torch and against the default
128-row kernel on the same inputs: forward (NT) and input-gradient (NN) forms, every epilogue the encoder uses, ragged M."""
import pytest
import torch

from tests.test_kernels_gpu import DEV, close, nat, rnd

import os

pytestmark = pytest.mark.gpu
T256 = 16384
# schedules of the 256-row kernel: plain ring, reads-first (bit 15), and - only on request, it has not run on hardware yet -
# the ping-pong schedule (bit 16): MMF_AMD_TEST_PINGPONG=1 python -m pytest tests/test_gemm256_gpu.py
SCHEDULES = [T256, T256 | 32768] + ([T256 | 65536] if os.environ.get("MMF_AMD_TEST_PINGPONG") == "1" else [])


def _same(C, C0, K):
    """Bit-identical to the 128-row kernel while both add the K-steps in the same order; from 24 K-steps on the default
    kernel splits K between wave groups (its KS = 2 layout), so the two differ by fp32 summation order only."""
    if K // 64 < 24:
        assert torch.equal(C, C0)
    else:
        close(C, C0.float(), 1e-2, 2e-2, "256-row vs 128-row kernel")


@pytest.mark.parametrize("flags", SCHEDULES)
@pytest.mark.parametrize("M,N,K", [(7296, 768, 768), (7296, 3072, 768), (1024, 768, 3072), (456, 2304, 768), (256, 128, 64), (300, 128, 128)])
def test_gemm256_forward_bias_matches_torch_and_the_default_kernel(M, N, K, flags):
    A = rnd(M, K, seed=1); B = rnd(N, K, seed=2, scale=0.05); bias = torch.randn(N, device=DEV)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); C0 = torch.empty_like(C)
    nat().gemm(A, B, C, M, N, K, K, K, N, bias=bias, debug_flags=flags)
    nat().gemm(A, B, C0, M, N, K, K, K, N, bias=bias)
    ref = A.float() @ B.float().t() + bias
    close(C, ref, 1e-2, 2e-2, "256-row tile forward")
    _same(C, C0, K)


def test_gemm256_gelu_residual_dropout_and_fp32_output():
    M, N, K = 1500, 768, 768
    A = rnd(M, K, seed=3); B = rnd(N, K, seed=4, scale=0.05); bias = torch.randn(N, device=DEV)
    for kw in (dict(act=1, U=torch.empty(M, N, dtype=torch.bfloat16, device=DEV)), dict(resid=rnd(M, N, seed=5), ldr=N, drop=nat().drop_cfg(0.1, 99))):
        C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); C0 = torch.empty_like(C)
        kw0 = dict(kw)
        if "U" in kw:
            kw0["U"] = torch.empty_like(kw["U"])
        nat().gemm(A, B, C, M, N, K, K, K, N, bias=bias, debug_flags=T256, **kw)
        nat().gemm(A, B, C0, M, N, K, K, K, N, bias=bias, **kw0)
        assert torch.equal(C, C0)
        if "U" in kw:
            assert torch.equal(kw["U"], kw0["U"])
    Cf = torch.empty(M, N, dtype=torch.float32, device=DEV)
    nat().gemm(A, B, Cf, M, N, K, K, K, N, debug_flags=T256)
    close(Cf, A.float() @ B.float().t(), 1e-4, 1e-3, "fp32 output")


@pytest.mark.parametrize("flags", SCHEDULES)
@pytest.mark.parametrize("M,N,K", [(7296, 768, 2304), (7296, 3072, 768), (520, 768, 768)])
def test_gemm256_dgrad_k_major_weight(M, N, K, flags):
    # dX [M, N] = dY [M, K] W [K, N]  (W k-major), with the gelu' multiply and the residual-gradient add of the FFN backward
    dY = rnd(M, K, seed=6); W = rnd(K, N, seed=7, scale=0.05)
    aux = rnd(M, N, seed=8); resid = rnd(M, N, seed=9)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); C0 = torch.empty_like(C)
    nat().gemm(dY, W, C, M, N, K, K, N, N, b_kmajor=True, act=2, aux=aux, resid=resid, ldr=N, debug_flags=flags)
    nat().gemm(dY, W, C0, M, N, K, K, N, N, b_kmajor=True, act=2, aux=aux, resid=resid, ldr=N)
    ref = (dY.float() @ W.float()) * aux.float() + resid.float()
    close(C, ref, 1e-2, 3e-2, "256-row tile dgrad")
    _same(C, C0, K)


def test_gemm256_repeated_launches_are_stable():
    """Race screen: the ring's RAW / WAR ordering must not depend on timing - 30 launches, identical results."""
    M, N, K = 7296, 2304, 768
    A = rnd(M, K, seed=11); B = rnd(N, K, seed=12, scale=0.05)
    C0 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    nat().gemm(A, B, C0, M, N, K, K, K, N)
    for flags in SCHEDULES:
        for _ in range(30):
            C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            nat().gemm(A, B, C, M, N, K, K, K, N, debug_flags=flags)
            assert torch.equal(C, C0), flags
