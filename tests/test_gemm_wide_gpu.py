"""The wide-tile forward GEMM (mmf_amd/csrc/gemm_wide.h: one 256x96 / 192x192 / 256x128 tile per CU, ping-pong wave groups over an
LDS-DMA ring) against fp32 torch and against the 128-row kernel on the same inputs — bit-identical, both add the K-steps in
the same order — for every epilogue the encoder uses, ragged M, short and long K, plus a repeated-launch race screen."""
import pytest
import torch

from tests.test_kernels_gpu import DEV, close, nat, rnd

pytestmark = pytest.mark.gpu
NO_WIDE = 1 << 17          # debug_flags: never a wide tile
NO_KSPLIT = 1 << 13        # keep the 128-row kernel on the plain K order
CONFIGS = {1: (256, 96), 2: (192, 192), 3: (256, 128)}


@pytest.fixture
def force_wide():
    def set_(cfg):
        nat().set_tunable(nat().TUN_GEMM_WIDE, cfg)
    yield set_
    nat().set_tunable(nat().TUN_GEMM_WIDE, 0)


@pytest.mark.parametrize("cfg", [1, 2, 3])
@pytest.mark.parametrize("M,K", [(7296, 768), (1500, 768), (512, 128), (3200, 3072), (2000, 192)])
def test_wide_forward_bias_matches_torch_and_the_128_row_kernel(cfg, M, K, force_wide):
    N = {1: 768, 2: 2304, 3: 3072}[cfg] if M > 2000 else {1: 192, 2: 384, 3: 512}[cfg]
    A = rnd(M, K, seed=1); B = rnd(N, K, seed=2, scale=0.05); bias = torch.randn(N, device=DEV)
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV); C0 = torch.empty_like(C)
    force_wide(cfg)
    nat().set_tunable(5, 1)          # MMF_TUN_GEMM_WIDE_KS = 1: the layout whose K order equals the 128-row kernel's
    try:
        nat().gemm(A, B, C, M, N, K, K, K, N, bias=bias)
    finally:
        nat().set_tunable(5, 0)
    nat().gemm(A, B, C0, M, N, K, K, K, N, bias=bias, debug_flags=NO_WIDE | NO_KSPLIT)
    close(C, A.float() @ B.float().t() + bias, 1e-2, 2e-2, "wide tile %s forward" % (CONFIGS[cfg],))
    assert torch.equal(C, C0)


@pytest.mark.parametrize("M,K", [(7296, 3072), (3200, 2304), (1500, 768), (300, 1536)])
def test_wide_256x96_k_split_layout_matches_torch_and_the_unsplit_layout(M, K, force_wide):
    """KS = 2 (round 3): the two ping-pong groups of the 256 x 96 tile multiply the two 32-deep halves of every K-step and the halves
    are summed in the epilogue's LDS stage: against fp32 torch, and against the unsplit layout up to the fp32 summation order (one
    bf16 rounding of the output), with every epilogue the fast path serves (bias + dropout + residual, act 2 + aux) and the generic one."""
    N = 768
    A = rnd(M, K, seed=11); B = rnd(N, K, seed=12, scale=0.05); bias = torch.randn(N, device=DEV)
    R = rnd(M, N, seed=13); aux = rnd(M, N, seed=14)
    force_wide(1)
    for kw, ref in ((dict(bias=bias), lambda r: r + bias), (dict(bias=bias, resid=R, ldr=N), lambda r: r + bias + R.float()),
                    (dict(act=2, aux=aux), lambda r: r * aux.float()),
                    (dict(bias=bias, act=1, U=torch.empty(M, N, dtype=torch.bfloat16, device=DEV)), lambda r: torch.nn.functional.gelu(r + bias))):
        outs = {}
        for ks in (1, 2):
            nat().set_tunable(5, ks)
            try:
                C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
                nat().gemm(A, B, C, M, N, K, K, K, N, **kw)
                outs[ks] = C
            finally:
                nat().set_tunable(5, 0)
        want = ref(A.float() @ B.float().t())
        close(outs[2], want, 1e-2, 2e-2 * float(want.abs().max()), "K-split layout vs torch %s" % sorted(kw))
        close(outs[2], outs[1], 1e-2, 1e-2 * float(want.abs().max()), "K-split vs unsplit layout %s" % sorted(kw))
    drop = nat().drop_cfg(0.1, 4242)
    outs = {}
    for ks in (1, 2):         # the same dropout mask under both layouts (hash of the element index)
        nat().set_tunable(5, ks)
        try:
            C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            nat().gemm(A, B, C, M, N, K, K, K, N, bias=bias, resid=R, ldr=N, drop=drop)
            outs[ks] = C
        finally:
            nat().set_tunable(5, 0)
    assert torch.equal((outs[1] == R), (outs[2] == R)) or float(((outs[1] == R) != (outs[2] == R)).float().mean()) < 1e-3
    close(outs[2], outs[1], 1e-2, 4e-2, "K-split vs unsplit layout under dropout")


@pytest.mark.parametrize("cfg", [1, 2, 3])
def test_wide_epilogues(cfg, force_wide):
    M, K = 1500, 768
    N = {1: 768, 2: 768, 3: 1024}[cfg]
    A = rnd(M, K, seed=3); B = rnd(N, K, seed=4, scale=0.05); bias = torch.randn(N, device=DEV)
    force_wide(cfg)
    cases = (dict(act=1, U=torch.empty(M, N, dtype=torch.bfloat16, device=DEV)),
             dict(resid=rnd(M, N, seed=5), ldr=N, drop=nat().drop_cfg(0.1, 99)),
             dict(act=2, aux=rnd(M, N, seed=6), resid=rnd(M, N, seed=7), ldr=N))
    for kw in cases:
        C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); C0 = torch.empty_like(C)
        kw0 = dict(kw)
        if "U" in kw:
            kw0["U"] = torch.empty_like(kw["U"])
        nat().gemm(A, B, C, M, N, K, K, K, N, bias=bias, **kw)
        nat().gemm(A, B, C0, M, N, K, K, K, N, bias=bias, debug_flags=NO_WIDE | NO_KSPLIT, **kw0)
        assert torch.equal(C, C0), kw.keys()
        if "U" in kw:
            assert torch.equal(kw["U"], kw0["U"])
    Cf = torch.empty(M, N, dtype=torch.float32, device=DEV)
    nat().gemm(A, B, Cf, M, N, K, K, K, N)
    close(Cf, A.float() @ B.float().t(), 1e-4, 1e-3, "fp32 output")


def test_wide_tile_is_what_the_dispatcher_picks_for_the_encoder_shapes(force_wide):
    """The auto choice at the VisualBERT VQA2 shapes must equal the forced wide configuration bit for bit (so the wide kernel is
    the one running) and differ in nothing from the 128-row kernel."""
    M = 7296
    for N, K, cfg in ((768, 768, 1), (2304, 768, 2), (3072, 768, 3), (768, 3072, 1)):
        A = rnd(M, K, seed=8); B = rnd(N, K, seed=9, scale=0.05)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); Cw = torch.empty_like(C)
        force_wide(0)
        nat().gemm(A, B, C, M, N, K, K, K, N)
        force_wide(cfg)
        nat().gemm(A, B, Cw, M, N, K, K, K, N)
        assert torch.equal(C, Cw)


@pytest.mark.parametrize("cfg,N", [(1, 768), (2, 2304), (3, 3072)])
def test_wide_repeated_launches_are_stable(cfg, N, force_wide):
    """Race screen: the ring's RAW / WAR ordering must not depend on timing — 30 launches on operands that other work evicts
    in between, identical results."""
    M, K = 7296, 768
    A = rnd(M, K, seed=11); B = rnd(N, K, seed=12, scale=0.05)
    C0 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    nat().gemm(A, B, C0, M, N, K, K, K, N, debug_flags=NO_WIDE | NO_KSPLIT)
    force_wide(cfg)
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    for i in range(30):
        if i % 3 == 0:
            junk.fill_(i)            # push the operands out of L2 every few launches
        C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        nat().gemm(A, B, C, M, N, K, K, K, N)
        assert torch.equal(C, C0), i
