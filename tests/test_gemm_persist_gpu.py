"""Parity of the persistent GEMM (mmf_amd/csrc/gemm_persist.h: one workgroup per CU walks several tiles, the epilogue of a tile runs under the
K-loop of the next) against an fp32 torch product of the same bf16 operands and against the one-tile-per-workgroup kernels it replaces —
every tile shape x every epilogue class, several tiles per workgroup, a ragged last row of tiles, dropout masks equal element for element.
Replaces the same reference calls as gemm.hip: nn.Linear forward / dgrad at mmf/modules/hf_layers.py:169-180,248,289-290."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
TUN_PERSIST = 18


def nat():
    from mmf_amd import _native
    return _native


@pytest.fixture(autouse=True)
def _reset_tunable():
    yield
    nat().lib().mmf_amd_set_tunable(TUN_PERSIST, 0)


def rnd(*shape, scale=1.0, dtype=torch.bfloat16, seed=0):
    g = torch.Generator(device="cpu").manual_seed(1000 + seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def run(tile, A, W, M, N, K, **kw):
    """One GEMM with the persistent kernel forced onto `tile` (1 / 2 / 3; -1: never), output buffers padded with canaries on both sides."""
    L = nat().lib()
    L.mmf_amd_set_tunable(TUN_PERSIST, tile)
    pad = 512 * N
    buf = torch.full((pad + M * N + pad,), 7.0, dtype=torch.bfloat16, device=DEV)
    C = buf[pad:pad + M * N].view(M, N)
    ubuf = None
    if kw.get("act") == 1:
        ubuf = torch.full((pad + M * N + pad,), 7.0, dtype=torch.bfloat16, device=DEV)
        kw = dict(kw, U=ubuf[pad:pad + M * N].view(M, N))
    nat().gemm(A, W, C, M, N, K, K, K, N, **kw)
    torch.cuda.synchronize()
    name = nat().gemm_last_kernel()
    for b in (buf, ubuf):
        if b is not None:       # nothing may be written outside [0, M) rows: the stores of rows >= M rely on the buffer descriptor's range check
            assert bool((b[:pad] == 7.0).all()) and bool((b[pad + M * N:] == 7.0).all()), "write outside the output (%s)" % name
    return C, (kw.get("U") if ubuf is not None else None), name


TILES = [(1, "256x96"), (2, "192x192"), (3, "256x128")]


@pytest.mark.parametrize("tile,tname", TILES)
@pytest.mark.parametrize("M,N,K", [(7296, 2304, 768), (1600, 1536, 1024), (4160, 768, 768)])
def test_persistent_plain_and_bias(tile, tname, M, N, K):
    A = rnd(M, K, scale=0.5); W = rnd(N, K, scale=0.05, seed=1); bias = rnd(N, dtype=torch.float32, seed=2)
    ref = A.float() @ W.float().t()
    for b in (None, bias):
        kw = {} if b is None else dict(bias=b)
        C, _, name = run(tile, A, W, M, N, K, **kw)
        assert name == "gemm_persist_kernel " + tname, name
        r = ref if b is None else ref + b
        err = float((C.float() - r).abs().max() / r.abs().max())
        assert err < 6e-3, (name, err)
        C0, _, name0 = run(-1, A, W, M, N, K, **kw)
        assert "persist" not in name0
        assert float((C.float() - C0.float()).abs().max()) <= 2.0 ** -6 * float(r.abs().max())      # same sums, another order: one bf16 ulp


@pytest.mark.parametrize("tile,tname", TILES)
def test_persistent_gelu_saves_the_derivative(tile, tname):
    M, N, K = 3200, 3072, 768
    A = rnd(M, K, scale=0.5); W = rnd(N, K, scale=0.05, seed=1); bias = rnd(N, dtype=torch.float32, seed=2)
    Hh, G, name = run(tile, A, W, M, N, K, bias=bias, act=1)
    assert name == "gemm_persist_kernel " + tname
    x = (A.float() @ W.float().t() + bias).requires_grad_(True)
    h = torch.nn.functional.gelu(x)
    g, = torch.autograd.grad(h.sum(), x)
    assert float((Hh.float() - h.detach()).abs().max()) < 2e-2 * float(h.detach().abs().max())
    assert float((G.float() - g).abs().max()) < 8e-3
    H0, G0, _ = run(-1, A, W, M, N, K, bias=bias, act=1)
    assert float((Hh.float() - H0.float()).abs().max()) <= 2.0 ** -6 * float(h.detach().abs().max())
    assert float((G.float() - G0.float()).abs().max()) <= 2.0 ** -6


@pytest.mark.parametrize("tile,tname", [(1, "256x96"), (3, "256x128")])     # (192 x 192 has no side-input form: out of registers)
@pytest.mark.parametrize("M,N,K", [(7296, 3072, 768), (2112, 768, 1536)])
def test_persistent_side_inputs_multiplier_residual_dropout(tile, tname, M, N, K):
    A = rnd(M, K, scale=0.5); W = rnd(N, K, scale=0.05, seed=1); bias = rnd(N, dtype=torch.float32, seed=2)
    R = rnd(M, N, seed=3); G = torch.rand(M, N, device=DEV).bfloat16()
    lin = A.float() @ W.float().t()
    # act 2: times the saved gelu'
    C, _, name = run(tile, A, W, M, N, K, act=2, aux=G)
    assert name == "gemm_persist_kernel " + tname
    r = lin * G.float()
    assert float((C.float() - r).abs().max() / r.abs().max()) < 6e-3
    # residual, with and without bias
    for kw in (dict(resid=R, ldr=N), dict(bias=bias, resid=R, ldr=N)):
        C, _, name = run(tile, A, W, M, N, K, **kw)
        assert name == "gemm_persist_kernel " + tname
        r = lin + R.float() + (bias if "bias" in kw else 0.0)
        assert float((C.float() - r).abs().max() / r.abs().max()) < 6e-3
    # hash dropout before the residual: the SAME mask as the one-tile kernels draw (a function of key and element index only)
    drop = nat().drop_cfg(0.1, 4242)
    Cd, _, _ = run(tile, A, W, M, N, K, bias=bias, resid=R, ldr=N, drop=drop)
    C0, _, name0 = run(-1, A, W, M, N, K, bias=bias, resid=R, ldr=N, drop=drop)
    assert "persist" not in name0
    assert float((Cd.float() - C0.float()).abs().max()) <= 2.0 ** -5 * float((lin.abs().max() * drop[2] + R.float().abs().max()))
    keep_frac = float(((Cd.float() - R.float()).abs() > 1e-3).float().mean())
    assert 0.86 < keep_frac < 0.92, keep_frac


def test_the_rule_takes_the_layer_shapes_it_was_measured_on():
    """Default dispatch: several tiles per workgroup and a short K-loop -> persistent; one round of tiles or a long K-loop -> the gemm_wide.h kernels."""
    M = 7296
    for N, K, want in [(2304, 768, True), (3072, 768, True), (768, 768, False), (768, 3072, False), (768, 2304, False)]:
        A = rnd(M, K); W = rnd(N, K, scale=0.05)
        _, _, name = run(0, A, W, M, N, K)
        assert ("persist" in name) == want, (N, K, name)


def test_persistent_is_deterministic():
    M, N, K = 7296, 3072, 768
    A = rnd(M, K, scale=0.5); W = rnd(N, K, scale=0.05, seed=1); bias = rnd(N, dtype=torch.float32, seed=2)
    C1, U1, _ = run(3, A, W, M, N, K, bias=bias, act=1)
    C1 = C1.clone(); U1 = U1.clone()
    for _ in range(3):
        C2, U2, _ = run(3, A, W, M, N, K, bias=bias, act=1)
        assert torch.equal(C1, C2) and torch.equal(U1, U2)
