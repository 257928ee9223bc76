"""Per-kernel parity tests (GPU): every HIP kernel, called through the C ABI, against a plain PyTorch
fp32 reference of the same op evaluated on the SAME bf16-rounded inputs.  Tolerances: bf16 outputs
carry one rounding (2^-9 relative) plus fp32 accumulation-order noise; fp32 outputs only the latter.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def nat():
    from mmf_amd import _native
    return _native


def rnd(*shape, scale=1.0, dtype=torch.bfloat16, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (hash(shape) & 0xFFFF) + 17)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def close(got, ref, rtol, atol, what=""):
    got = got.float(); ref = ref.float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError("%s: %d/%d elements off, max err %.4g (ref scale %.4g), first bad %s got %.5g ref %.5g" % (
            what, int(bad.sum()), bad.numel(), float(err.max()), float(ref.abs().max()), idx,
            float(got[tuple(idx)]), float(ref[tuple(idx)])))


# ---------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (384, 768, 768), (100, 200, 72), (32, 3129, 768), (7296, 768, 768)])
def test_gemm_forward_bias(M, N, K):
    A = rnd(M, K); B = rnd(N, K); bias = rnd(N, dtype=torch.float32)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    nat().gemm(A, B, C, M, N, K, K, K, N, bias=bias)
    ref = A.float() @ B.float().t() + bias
    close(C, ref, 1e-2, 2e-2 * math.sqrt(K) / 8, "fwd bf16")
    # fp32 output (scores head), ragged ldc
    C32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    nat().gemm(A, B, C32, M, N, K, K, K, N, bias=bias)
    close(C32, ref, 1e-4, 1e-3 * math.sqrt(K) / 8, "fwd f32")


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (384, 768, 3072), (100, 72, 200), (32, 768, 3129)])
def test_gemm_dgrad_b_kmajor(M, N, K):
    # dX[M,N] = dY[M,K] W[K,N]  (W row index is the reduction index)
    ldk = (K + 7) // 8 * 8
    dY = torch.zeros(M, ldk, dtype=torch.bfloat16, device=DEV)
    dY[:, :K] = rnd(M, K)
    W = rnd(K, N) if N % 8 == 0 else None
    if W is None:
        pytest.skip("ldb must be a multiple of 8")
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    nat().gemm(dY, W, out, M, N, K, ldk, N, N, b_kmajor=True)
    ref = dY[:, :K].float() @ W.float()
    close(out, ref, 1e-2, 2e-2 * math.sqrt(K) / 8, "dgrad")


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (768, 3072, 7296), (768, 768, 320), (3129, 768, 32), (200, 136, 100)])
def test_gemm_wgrad_both_kmajor(M, N, K):
    # dW[M,N] = dY^T X : A = dY [K, M] k-major, B = X [K, N] k-major
    ldm = (M + 7) // 8 * 8
    dY = torch.zeros(K, ldm, dtype=torch.bfloat16, device=DEV); dY[:, :M] = rnd(K, M)
    X = rnd(K, N)
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    nat().gemm(dY, X, out, M, N, K, ldm, N, N, a_kmajor=True, b_kmajor=True)
    ref = dY[:, :M].float().t() @ X.float()
    close(out, ref, 1e-4, 2e-3 * math.sqrt(K) / 8, "wgrad")
    # accumulate (beta = 1)
    nat().gemm(dY, X, out, M, N, K, ldm, N, N, a_kmajor=True, b_kmajor=True, beta=1.0)
    close(out, 2 * ref, 1e-4, 4e-3 * math.sqrt(K) / 8, "wgrad accumulate")


def test_gemm_visual_projection_epilogue():
    # embeddings.py:352-367: projection(feat) + position_embeddings_visual[0] + token_type_embeddings_visual[type]
    B, R, T, H, D = 3, 100, 28, 768, 2048
    S = T + R
    feat = rnd(B * R, D, dtype=torch.float32).abs()
    W = rnd(H, D, scale=0.02); bias = rnd(H, dtype=torch.float32)
    typ = rnd(2, H, dtype=torch.float32); pos0 = rnd(H, dtype=torch.float32)
    vt = torch.randint(0, 2, (B * R,), device=DEV)
    y = torch.zeros(B * S, H, dtype=torch.bfloat16, device=DEV)
    nat().gemm(feat, W, y, B * R, H, D, D, D, H, bias=bias, coladd=pos0, rowtab=typ, rowidx=vt, rowtab_ld=H, grp=(R, T, T))
    ref = feat.to(torch.bfloat16).float() @ W.float().t() + bias + pos0 + typ[vt]
    got = y.view(B, S, H)[:, T:, :].reshape(B * R, H)
    close(got, ref, 1e-2, 3e-2, "visual projection")
    assert float(y.view(B, S, H)[:, :T].abs().max()) == 0.0
    # wgrad with fp32 k-major B: dW[H, D] = dV^T feat
    dV = rnd(B * R, H)
    dW = torch.empty(H, D, dtype=torch.float32, device=DEV)
    nat().gemm(dV, feat, dW, H, D, B * R, H, D, D, a_kmajor=True, b_kmajor=True)
    close(dW, dV.float().t() @ feat.to(torch.bfloat16).float(), 1e-4, 2e-2, "visual wgrad")


@pytest.mark.parametrize("Nout,Kin,T", [(768, 768, 7296), (2304, 768, 7296), (3072, 768, 2048), (200, 136, 4096), (768, 3072, 7296)])
def test_gemm_wgrad_carries_the_bias_gradient(Nout, Kin, T):
    """Weight-gradient form with `rowsum_out`: the same launch returns dW = dY^T X and db = column sums of dY (ones-operand
    MFMA in the first column of workgroups, extra slab column, summed by the slab reduction)."""
    assert nat().gemm_rowsum_supported(Nout, Kin, T)
    dY = rnd(T, Nout); X = rnd(T, Kin)
    dW = torch.empty(Nout, Kin, dtype=torch.float32, device=DEV); db = torch.full((Nout,), float("nan"), device=DEV)
    nat().gemm(dY, X, dW, Nout, Kin, T, Nout, Kin, Kin, a_kmajor=True, b_kmajor=True, rowsum_out=db)
    close(dW, dY.float().t() @ X.float(), 1e-4, 2e-3 * math.sqrt(T) / 8, "dW with fused bias gradient")
    close(db, dY.float().sum(0), 1e-5, 1e-3, "fused bias gradient")
    # and dW is what the plain weight-gradient launch gives
    dW2 = torch.empty_like(dW)
    nat().gemm(dY, X, dW2, Nout, Kin, T, Nout, Kin, Kin, a_kmajor=True, b_kmajor=True, debug_flags=8192)
    close(dW, dW2, 1e-6, 1e-5, "dW with / without the fused bias gradient")
    # a reduction too short to split K: the row sums are written directly by the one workgroup per tile
    db2 = torch.full((Nout,), float("nan"), device=DEV)
    nat().gemm(dY[:64], X[:64], dW, Nout, Kin, 64, Nout, Kin, Kin, a_kmajor=True, b_kmajor=True, rowsum_out=db2)
    close(dW, dY[:64].float().t() @ X[:64].float(), 1e-4, 1e-3, "dW, unsplit")
    close(db2, dY[:64].float().sum(0), 1e-5, 1e-3, "bias gradient, unsplit")


@pytest.mark.parametrize("T", [7296, 456, 64])
def test_gemm_grouped_layer_weight_gradients(T):
    """The four weight gradients of a transformer layer (+ two bias gradients) as ONE grouped launch, no split-K: each against
    fp32 torch and against its own separate launch (split-K there, so equal up to fp32 summation order)."""
    H, I = 768, 3072
    du = rnd(T, I, seed=1); a_out = rnd(T, H, seed=2); dlin2 = rnd(T, H, seed=3); hh = rnd(T, I, seed=4)
    dqkv = rnd(T, 3 * H, seed=5); x = rnd(T, H, seed=6); dlin1 = rnd(T, H, seed=7); ctxt = rnd(T, H, seed=8)
    specs = [(du, a_out, I, H, True), (dlin2, hh, H, I, False), (dqkv, x, 3 * H, H, True), (dlin1, ctxt, H, H, False)]
    probs, outs = [], []
    for dy, xx, N, K, want_db in specs:
        dw = torch.full((N, K), float("nan"), device=DEV); db = torch.full((N,), float("nan"), device=DEV) if want_db else None
        probs.append(dict(A=dy, B=xx, C_out=dw, M=N, N=K, K=T, lda=N, ldb=K, ldc=K, a_kmajor=True, b_kmajor=True, rowsum_out=db))
        outs.append((dw, db))
    nat().gemm_grouped(probs)
    for (dy, xx, N, K, want_db), (dw, db) in zip(specs, outs):
        ref = dy.float().t() @ xx.float()
        close(dw, ref, 1e-4, 2e-3 * math.sqrt(T) / 8, "grouped dW %dx%d" % (N, K))
        sep = torch.empty_like(dw)
        nat().gemm(dy, xx, sep, N, K, T, N, K, K, a_kmajor=True, b_kmajor=True)
        close(dw, sep, 1e-5, 2e-4 * math.sqrt(T) / 8, "grouped vs separate launch")
        if want_db:
            close(db, dy.float().sum(0), 1e-5, 1e-3 * math.sqrt(T) / 8, "grouped bias gradient")


@pytest.mark.parametrize("T,H,p,rides", [(7296, 768, 0.1, True), (3648, 768, 0.0, True), (8704, 768, 0.1, True), (456, 768, 0.1, False), (4096, 1024, 0.1, False)])
def test_layernorm_backward_rides_the_grouped_weight_gradients_bit_for_bit(T, H, p, rides):
    """mmf_gemm_bf16_grouped_ln: a layer's four weight gradients and the deferred LayerNorm backward of the layer below in ONE launch (the LayerNorm on the CUs
    the 216 gradient tiles leave idle) against the two launches one after the other: every output the same bits — dW, db, dx, the dropout-masked dlin, the
    column-sum partials.  Shapes that do not ride (token count not 64-aligned: no wide tiles; H = 1024: not the rider's form) take the two launches inside
    the entry point.  8704 rows: more than the LayerNorm's 512 blocks hold two rows each, so its row loop runs twice."""
    I = 4 * H
    du = rnd(T, I, seed=1); a_out = rnd(T, H, seed=2); dlin2 = rnd(T, H, seed=3); hh = rnd(T, I, seed=4)
    dqkv = rnd(T, 3 * H, seed=5); x = rnd(T, H, seed=6); dlin1 = rnd(T, H, seed=7); ctxt = rnd(T, H, seed=8)
    specs = [(du, a_out, I, H), (dlin2, hh, H, I), (dqkv, x, 3 * H, H), (dlin1, ctxt, H, H)]
    dy = rnd(T, H, seed=9); y = rnd(T, H, seed=10)
    mean = rnd(T, dtype=torch.float32, seed=11); rstd = rnd(T, dtype=torch.float32, seed=12).abs() + 0.5; gamma = rnd(H, dtype=torch.float32, seed=13)
    drop = nat().drop_cfg(p, 4242, torch.tensor([7], dtype=torch.int32, device=DEV)) if p else nat().NO_DROP
    res = []
    for ride in (False, True):
        probs = []
        for a, b, N, K in specs:
            probs.append(dict(A=a, B=b, C_out=torch.full((N, K), float("nan"), device=DEV), M=N, N=K, K=T, lda=N, ldb=K, ldc=K, a_kmajor=True, b_kmajor=True,
                              rowsum_out=torch.full((N,), float("nan"), device=DEV)))
        dx = torch.full((T, H), float("nan"), dtype=torch.bfloat16, device=DEV); dlin = torch.full_like(dx, float("nan")) if p else None
        ws = torch.zeros(nat().layernorm_bwd_ws_floats(H), device=DEV)
        if ride:
            nat().gemm_grouped_ln(probs, dy, y, mean, rstd, gamma, dx, dlin, drop, ws, T, H)
            assert ("grouped_ln" in nat().gemm_last_kernel()) == rides
        else:
            nat().gemm_grouped(probs)
            nat().layernorm_bwd(dy, y, mean, rstd, gamma, dx, dlin, drop, None, None, None, 0, ws, T, H)
        torch.cuda.synchronize()
        res.append([q["C_out"] for q in probs] + [q["rowsum_out"] for q in probs] + [dx, ws] + ([dlin] if p else []))
    for a, b in zip(*res):
        assert not bool(torch.isnan(a.float()).any())
        assert torch.equal(a, b)
    # the partials reduce to the parameter gradients of the plain (non-deferred) backward
    dg = torch.empty(H, device=DEV); db = torch.empty(H, device=DEV)
    nat().layernorm_bwd_reduce_multi([(res[1][9], T, H, dg, db)])
    dg0 = torch.empty(H, device=DEV); db0 = torch.empty(H, device=DEV); dx0 = torch.empty_like(res[1][8]); dl0 = torch.empty_like(dx0) if p else None
    nat().layernorm_bwd(dy, y, mean, rstd, gamma, dx0, dl0, drop, dg0, db0, None, 0, torch.zeros_like(res[1][9]), T, H)
    assert torch.equal(dx0, res[1][8])
    close(dg, dg0, 1e-4, 1e-2, "dgamma from the rider's partials"); close(db, db0, 1e-4, 1e-2, "dbeta from the rider's partials")


def test_gemm_grouped_forward_problems_keep_their_epilogues():
    """Grouping is layout-generic: two forward GEMMs of different shapes, each with its own bias / residual, in one launch;
    results are bit-identical to the separate launches (same tiles, same K order)."""
    A1 = rnd(300, 768, seed=1); W1 = rnd(384, 768, seed=2, scale=0.05); b1 = rnd(384, dtype=torch.float32, seed=3)
    A2 = rnd(1024, 256, seed=4); W2 = rnd(768, 256, seed=5, scale=0.05); b2 = rnd(768, dtype=torch.float32, seed=6); R2 = rnd(1024, 768, seed=7)
    C1 = torch.empty(300, 384, dtype=torch.bfloat16, device=DEV); C2 = torch.empty(1024, 768, dtype=torch.bfloat16, device=DEV)
    nat().gemm_grouped([dict(A=A1, B=W1, C_out=C1, M=300, N=384, K=768, lda=768, ldb=768, ldc=384, bias=b1),
                        dict(A=A2, B=W2, C_out=C2, M=1024, N=768, K=256, lda=256, ldb=256, ldc=768, bias=b2, resid=R2, ldr=768)])
    S1 = torch.empty_like(C1); S2 = torch.empty_like(C2)
    nat().gemm(A1, W1, S1, 300, 384, 768, 768, 768, 384, bias=b1, debug_flags=512 | 8192)
    nat().gemm(A2, W2, S2, 1024, 768, 256, 256, 256, 768, bias=b2, resid=R2, ldr=768, debug_flags=512 | 8192)
    assert torch.equal(C1, S1) and torch.equal(C2, S2)
    with pytest.raises(Exception):      # mixed operand layouts are refused
        nat().gemm_grouped([dict(A=A1, B=W1, C_out=C1, M=300, N=384, K=768, lda=768, ldb=768, ldc=384),
                            dict(A=A2, B=rnd(256, 768, seed=9), C_out=C2, M=1024, N=768, K=256, lda=256, ldb=768, ldc=768, b_kmajor=True)])


def test_gemm_timeline_probe_records_every_workgroup():
    M, N, K = 1024, 768, 768
    A = rnd(M, K); B = rnd(N, K); C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    buf = torch.zeros(8 * (1 + 256), dtype=torch.int64, device=DEV)
    nat().gemm_set_probe(buf)
    try:
        nat().gemm(A, B, C, M, N, K, K, K, N, debug_flags=512 | (1 << 17))      # 128 x 128 tiles of the 128-row kernel
        torch.cuda.synchronize()
    finally:
        nat().gemm_set_probe(None)
    n = int(buf[0].item())
    assert n == (M // 128) * (N // 128)
    rec = buf[8:8 * (1 + n)].view(n, 8)
    assert bool((rec[:, 3] >= rec[:, 2]).all()) and bool((rec[:, 6] >= rec[:, 4]).all())
    close(C, A.float() @ B.float().t(), 1e-2, 5e-2, "probed launch still computes")


@pytest.mark.parametrize("N", [768, 2304, 3072, 384])
def test_gemm_tile_variants_agree(N):
    """8-wave 128x128 / 128x96 tiles (production), the 4-wave 128x128 form and the K-split wave layout compute the same thing."""
    M, K = 1024, 768
    A = rnd(M, K); B = rnd(N, K); bias = rnd(N, dtype=torch.float32); R = rnd(M, N)
    C1 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); C2 = torch.empty_like(C1)
    nat().gemm(A, B, C1, M, N, K, K, K, N, bias=bias, resid=R, ldr=N)
    nat().gemm(A, B, C2, M, N, K, K, K, N, bias=bias, resid=R, ldr=N, debug_flags=256)
    assert torch.equal(C1, C2)
    Bk = rnd(K, N)
    nat().gemm(A, Bk, C1, M, N, K, K, N, N, b_kmajor=True)
    nat().gemm(A, Bk, C2, M, N, K, K, N, N, b_kmajor=True, debug_flags=256)
    assert torch.equal(C1, C2)
    Ak = rnd(K, M)
    W1 = torch.empty(M, N, dtype=torch.float32, device=DEV); W2 = torch.empty_like(W1)
    nat().gemm(Ak, Bk, W1, M, N, K, M, N, N, a_kmajor=True, b_kmajor=True)
    nat().gemm(Ak, Bk, W2, M, N, K, M, N, N, a_kmajor=True, b_kmajor=True, debug_flags=256)
    close(W1, W2, 1e-5, 1e-4, "wgrad 8-wave vs 4-wave")
    C3 = torch.empty_like(C1); W3 = torch.empty_like(W1)
    # K-split wave layout (bit 12 forces it, bit 13 forbids it): same sums in a different order
    nat().gemm(A, B, C3, M, N, K, K, K, N, bias=bias, resid=R, ldr=N, debug_flags=4096)
    nat().gemm(A, B, C2, M, N, K, K, K, N, bias=bias, resid=R, ldr=N, debug_flags=8192)
    close(C3, C2, 1e-2, 1e-2, "fwd K-split layout")
    nat().gemm(A, Bk, C3, M, N, K, K, N, N, b_kmajor=True, debug_flags=4096)
    nat().gemm(A, Bk, C2, M, N, K, K, N, N, b_kmajor=True, debug_flags=8192)
    close(C3, C2, 1e-2, 1e-2, "dgrad K-split layout")
    nat().gemm(Ak, Bk, W3, M, N, K, M, N, N, a_kmajor=True, b_kmajor=True, debug_flags=4096)
    nat().gemm(Ak, Bk, W2, M, N, K, M, N, N, a_kmajor=True, b_kmajor=True, debug_flags=8192)
    close(W3, W2, 1e-5, 1e-4, "wgrad K-split layout")


def test_gemm_gelu_and_dgelu_epilogues():
    M, N, K = 256, 512, 256
    A = rnd(M, K); W = rnd(N, K, scale=0.05); bias = rnd(N, dtype=torch.float32)
    G = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); Hh = torch.empty_like(G)
    nat().gemm(A, W, Hh, M, N, K, K, K, N, bias=bias, act=1, U=G)
    u_ref = (A.float() @ W.float().t() + bias).requires_grad_(True)
    h_ref = torch.nn.functional.gelu(u_ref)
    h_ref.sum().backward()
    close(Hh, h_ref.detach(), 1e-2, 2e-2, "gelu")
    close(G, u_ref.grad, 1e-2, 1e-2, "gelu' saved by the forward epilogue")
    # dgelu: out = (dY Wd) * G
    dY = rnd(M, 128); Wd = rnd(128, N, scale=0.1)
    dU = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    nat().gemm(dY, Wd, dU, M, N, 128, 128, N, N, b_kmajor=True, act=2, aux=G)
    close(dU, (dY.float() @ Wd.float()) * G.float(), 1e-2, 2e-2, "dgelu")
    dU2 = torch.empty_like(dU)
    dH = rnd(M, N)
    nat().gelu_bwd(dH, G, dU2)
    close(dU2, dH.float() * G.float(), 1e-2, 1e-3, "standalone gelu backward")


def test_fast_erf_gelu_is_accurate_in_fp32():
    """gelu / gelu' (Abramowitz-Stegun erf) against torch's exact-erf GELU through an fp32-output GEMM with K padding."""
    M, N, K = 128, 128, 64
    x = torch.linspace(-9.0, 9.0, M * N, device=DEV).view(M, N)
    eye = torch.zeros(N, K, dtype=torch.bfloat16, device=DEV)
    A = torch.zeros(M, K, dtype=torch.bfloat16, device=DEV)
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    G = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    nat().gemm(A, eye, out, M, N, K, K, K, N, rowtab=x.contiguous(), rowidx=torch.arange(M, device=DEV), rowtab_ld=N, act=1, U=G)
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.functional.gelu(xr)
    ref.sum().backward()
    assert float((out - ref.detach()).abs().max()) < 2e-6
    assert float((G.float() - xr.grad).abs().max()) < 6e-3   # bf16 storage of values in [-0.13, 1.13]


def test_gemm_residual_and_dropout_epilogue():
    M, N, K = 384, 768, 256
    A = rnd(M, K); W = rnd(N, K, scale=0.05); bias = rnd(N, dtype=torch.float32); R = rnd(M, N)
    Y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    nat().gemm(A, W, Y, M, N, K, K, K, N, bias=bias, resid=R, ldr=N)
    lin = A.float() @ W.float().t() + bias
    close(Y, lin + R.float(), 1e-2, 2e-2, "residual")
    drop = nat().drop_cfg(0.1, 12345)
    Yd = torch.empty_like(Y)
    nat().gemm(A, W, Yd, M, N, K, K, K, N, bias=bias, resid=R, ldr=N, drop=drop)
    d = Yd.float() - R.float()
    kept = (d - lin * drop[2]).abs() <= 2e-2 + 2e-2 * lin.abs()
    dropped = d.abs() <= 2e-2
    assert bool((kept | dropped).all())
    frac = float((~kept & dropped).float().mean())
    assert 0.08 < frac < 0.12, frac
    # LayerNorm-backward's dropout mask must be the same mask (same key / index convention)
    mean = torch.zeros(M, device=DEV); rstd = torch.ones(M, device=DEV); gamma = torch.ones(N, device=DEV)
    dy = rnd(M, N); x = rnd(M, N)
    dx = torch.empty_like(dy); dlin = torch.empty_like(dy)
    ws = torch.empty(nat().layernorm_bwd_ws_floats(N), device=DEV)
    nat().layernorm_bwd(dy, x, mean, rstd, gamma, dx, dlin, drop, None, None, None, 0, ws, M, N)
    sel = (lin.abs() > 0.1) & (dx.float().abs() > 0)     # elements where both masks are observable
    mask_ln = dlin.float().abs() > 0
    mask_gemm = d.abs() > 0.05
    assert int(sel.sum()) > 0.5 * sel.numel()
    assert torch.equal(mask_ln[sel], mask_gemm[sel])


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
def attn_ref(q, k, v, mask, scale, pmask=None, pscale=1.0):
    # q,k,v [B, heads, S, 64] fp32; mask [B, Sk] additive
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if mask is not None:
        s = s + mask[:, None, None, :]
    p = torch.softmax(s, dim=-1)
    lse = torch.logsumexp(s, dim=-1)
    if pmask is not None:
        p = p * pmask * pscale
    return torch.matmul(p, v), lse


def split_heads(x, B, S, heads):
    return x.view(B, S, heads, 64).permute(0, 2, 1, 3).float()


@pytest.mark.parametrize("B,heads,S", [(2, 3, 228), (1, 2, 100), (1, 1, 256), (2, 2, 33), (1, 12, 128)])
def test_attention_forward_backward(B, heads, S):
    H = heads * 64
    qkv = rnd(B * S, 3 * H, scale=1.0)
    mbin = (torch.rand(B, S, device=DEV) > 0.2).long()
    mbin[:, 0] = 1
    mask = torch.empty(B, S, device=DEV)
    nat().make_additive_mask(mbin, mask)
    assert torch.equal(mask, (1.0 - mbin.float()) * -10000.0)
    ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, heads, S, device=DEV)
    scale = 1.0 / math.sqrt(64)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    nat().attention_fwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, heads, S, S, scale)
    qf = split_heads(qkv[:, :H].contiguous(), B, S, heads).requires_grad_(True)
    kf = split_heads(qkv[:, H:2 * H].contiguous(), B, S, heads).requires_grad_(True)
    vf = split_heads(qkv[:, 2 * H:].contiguous(), B, S, heads).requires_grad_(True)
    o_ref, lse_ref = attn_ref(qf, kf, vf, mask, scale)
    got = split_heads(ctx, B, S, heads)
    close(got, o_ref, 2e-2, 2e-2, "attention ctx")
    close(lse, lse_ref, 1e-4, 2e-3, "lse")
    # backward
    dctx = rnd(B * S, H)
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty(B, heads, S, device=DEV)
    nat().attention_bwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, heads, S, S, scale, dctx,
                        dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:], delta)
    o_ref.backward(split_heads(dctx, B, S, heads))
    for name, got_, ref_ in (("dq", dqkv[:, :H], qf.grad), ("dk", dqkv[:, H:2 * H], kf.grad), ("dv", dqkv[:, 2 * H:], vf.grad)):
        g = split_heads(got_.contiguous(), B, S, heads)
        close(g, ref_, 3e-2, 3e-2 * float(ref_.abs().max()), name)


@pytest.mark.parametrize("B,heads,S,tail,p", [(32, 12, 228, 0, 0.1), (3, 4, 200, 0, 0.0), (2, 2, 256, 0, 0.1), (2, 3, 182, 12, 0.1), (1, 1, 129, 0, 0.0)])
def test_attention_one_round_forward_is_bit_identical_to_the_two_workgroup_form(B, heads, S, tail, p):
    """Round 3: head_dim 64 with more than 128 queries runs as ONE 8-wave workgroup per (batch, head) that computes the scores twice
    (row maxima, then exponentials) instead of holding them in 128 registers — two workgroups per CU, one round.  Same arithmetic in
    the same order as the two-workgroups-per-head kernel (MMF_TUN_ALT_FORMS bit 1): context (bf16 and fp32 copy) and log-sum-exp
    bit for bit, with ragged masks, dropout and the prefix-LM tail."""
    H = heads * 64
    qkv = rnd(B * S, 3 * H, seed=21)
    mask = torch.zeros(B, S, device=DEV)
    for b in range(B):
        mask[b, S - 1 - 7 * (b % 5):] = -10000.0
    drop = nat().drop_cfg(p, 777) if p > 0 else nat().NO_DROP
    outs = {}
    for old in (1, 0):
        nat().set_tunable(nat().TUN_ALT_FORMS, 2 * old)
        try:
            ctx = torch.full((B * S, H), float("nan"), dtype=torch.bfloat16, device=DEV)
            c32 = torch.full((B * S, H), float("nan"), dtype=torch.float32, device=DEV)
            lse = torch.full((B, heads, S), float("nan"), device=DEV)
            nat().attention_fwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, heads, S, S, 0.125, drop, ctx_f32=c32,
                                causal_tail=tail)
            outs[old] = (ctx, c32, lse)
        finally:
            nat().set_tunable(nat().TUN_ALT_FORMS, 0)
    for a, b_ in zip(outs[0], outs[1]):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a, b_)


@pytest.mark.parametrize("S,mode", [(228, "key"), (160, "key"), (256, "tail"), (200, "query"), (129, "key"), (128, "key"), (100, "tail"), (40, "query"), (7, "key")])
def test_attention_keep_bit_table_replays_the_forwards_dropout_decisions(S, mode):
    """mmf_attn_desc.keep_bits: the forward writes its probability-dropout decisions as a bit table while it draws them, the one-pass backward reads one
    word per lane and key tile instead of hashing every probability again.  Same decisions: forward output and all three gradients are bit-identical to
    the hashing path, for the key mask, M4C's causal tail and a materialised per-query mask; shapes whose kernels take no table report 0 words."""
    B, heads, d = 3, 4, 64
    H = heads * d
    qkv = rnd(B * S, 3 * H, scale=0.5, seed=S)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    keym = torch.zeros(B, S, device=DEV); keym[:, S - 7:] = -10000.0
    tail = 12 if mode == "tail" else 0
    mask = keym
    if mode == "query":
        mask = (keym[:, None, :] + torch.where(torch.rand(B, S, S, device=DEV) < 0.1, -10000.0, 0.0)).contiguous()
    drop = nat().drop_cfg(0.1, 424242)
    dctx = rnd(B * S, H, scale=0.1, seed=3)
    words = nat().attention_keep_bits_words(B, heads, S, S, d)
    assert words == B * heads * ((S + 31) // 32) ** 2 * 32
    outs = []
    for use in (False, True):
        kb = torch.full((words,), -1 if use else 0, dtype=torch.int32, device=DEV) if use else None
        ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV); lse = torch.empty(B, heads, S, device=DEV); o32 = torch.empty(B * S, H, device=DEV)
        nat().attention_fwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, heads, S, S, 0.125, drop, ctx_f32=o32, causal_tail=tail, keep_bits=kb)
        dqkv = torch.zeros_like(qkv); delta = torch.empty(B, heads, S, device=DEV)
        nat().attention_bwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, heads, S, S, 0.125, dctx, dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:],
                            delta, drop, ctx_f32=o32, causal_tail=tail, keep_bits=kb)
        outs.append((ctx, lse, dqkv, kb))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], outs[1][2])
    # the table: ~10 % of the bits of valid (query, key) pairs are zero
    kb = outs[1][3].view(B * heads, (S + 31) // 32, (S + 31) // 32, 32)
    word = kb[:, 0, 0, :].contiguous().view(-1)       # tile (0, 0): 32 x 32 valid pairs per word set
    zeros = sum(int(((word >> b) & 1 == 0).sum()) for b in range(32))
    assert 0.07 < zeros / (word.numel() * 32) < 0.13
    # shapes without a table
    for (sq, sk, hd) in ((300, 300, 64), (100, 257, 64), (228, 228, 128), (129, 64, 128)):
        assert nat().attention_keep_bits_words(B, heads, sq, sk, hd) == 0
    with pytest.raises(nat().NativeLibraryError):       # a table where the kernels take none is an error, not silently ignored
        S2 = 288
        x = rnd(B * S2, 3 * H)
        nat().attention_fwd(x[:, :H], x[:, H:2 * H], x[:, 2 * H:], 3 * H, 3 * H, 3 * H, None, torch.empty(B * S2, H, dtype=torch.bfloat16, device=DEV), H,
                            torch.empty(B, heads, S2, device=DEV), B, heads, S2, S2, 0.125, drop, keep_bits=torch.zeros(1024, dtype=torch.int32, device=DEV))


@pytest.mark.parametrize("Sq,Sk,d", [(128, 101, 128), (101, 128, 128), (36, 20, 128), (128, 101, 64), (200, 64, 64), (100, 228, 64), (64, 256, 64), (256, 130, 64)])
def test_attention_keep_bit_table_cross_attention_and_head_dim_128(Sq, Sk, d):
    """The same for q and k / v from different sequences (ViLBERT's co-attention, vilbert.py:388-475) and head_dim 128 (its visual stream)."""
    B, heads = 2, 3
    H = heads * d
    q = rnd(B * Sq, H, scale=0.5, seed=Sq); kv = rnd(B * Sk, 2 * H, scale=0.5, seed=Sk + 1)
    mask = torch.zeros(B, Sk, device=DEV); mask[:, Sk - 3:] = -10000.0
    drop = nat().drop_cfg(0.1, 777)
    dctx = rnd(B * Sq, H, scale=0.1, seed=9)
    words = nat().attention_keep_bits_words(B, heads, Sq, Sk, d)
    assert words == B * heads * ((Sq + 31) // 32) * ((Sk + 31) // 32) * 32
    outs = []
    for use in (False, True):
        kb = torch.zeros(words, dtype=torch.int32, device=DEV) if use else None
        ctx = torch.empty(B * Sq, H, dtype=torch.bfloat16, device=DEV); lse = torch.empty(B, heads, Sq, device=DEV)
        nat().attention_fwd(q, kv, kv[:, H:], H, 2 * H, 2 * H, mask, ctx, H, lse, B, heads, Sq, Sk, 1.0 / math.sqrt(d), drop, head_dim=d, keep_bits=kb)
        dq = torch.zeros_like(q); dkv = torch.zeros_like(kv); delta = torch.empty(B, heads, Sq, device=DEV)
        nat().attention_bwd(q, kv, kv[:, H:], H, 2 * H, 2 * H, mask, ctx, H, lse, B, heads, Sq, Sk, 1.0 / math.sqrt(d), dctx, dq, dkv, dkv[:, H:], delta, drop,
                            head_dim=d, keep_bits=kb)
        outs.append((ctx, dq, dkv))
    for a_, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("Sq,Sk,d,mode", [(228, 228, 64, "key"), (256, 256, 64, "tail"), (200, 200, 64, "query"), (160, 160, 64, "key"), (128, 128, 64, "key"),
                                          (100, 100, 64, "tail"), (40, 40, 64, "query"), (7, 7, 64, "key"), (128, 101, 128, "key"), (36, 20, 128, "key"),
                                          (200, 64, 64, "key"), (100, 228, 64, "key")])
def test_attention_dropout_decisions_drawn_ahead_of_the_forward(Sq, Sk, d, mode):
    """mmf_attention_draw_keep_bits: the decisions depend on (key, seed word, element index) only, so ONE launch draws them for several attention calls
    ahead of the kernels.  The key-major table equals what the forward writes while hashing, bit for bit; a forward that reads
    `keep_lanes` returns the same context / log-sum-exp bits as the hashing forward, for every kernel form (8-wave and 4-wave head_dim 64, head_dim 128,
    cross attention, causal tail, per-query mask), with a device seed word as a replayed graph uses it."""
    B, heads = 3, 4
    H = heads * d
    q = rnd(B * Sq, H, scale=0.5, seed=Sq); kv = rnd(B * Sk, 2 * H, scale=0.5, seed=Sk + 1)
    keym = torch.zeros(B, Sk, device=DEV); keym[:, Sk - 3:] = -10000.0
    tail = min(12, Sk - 1) if mode == "tail" else 0
    mask = keym
    if mode == "query":
        mask = (keym[:, None, :] + torch.where(torch.rand(B, Sq, Sk, device=DEV) < 0.1, -10000.0, 0.0)).contiguous()
    seed = torch.tensor([12345], dtype=torch.int32, device=DEV)
    drops = [nat().drop_cfg(0.1, 424242 + i, seed) for i in range(3)]
    wb, wl = nat().attention_keep_bits_words(B, heads, Sq, Sk, d), nat().attention_keep_lanes_words(B, heads, Sq, Sk, d)
    assert wb > 0 and wl == B * heads * ((Sq + 31) // 32) * 256
    scale = 1.0 / math.sqrt(d)

    def fwd(drop, kb, kl):
        ctx = torch.empty(B * Sq, H, dtype=torch.bfloat16, device=DEV); lse = torch.empty(B, heads, Sq, device=DEV); o32 = torch.empty(B * Sq, H, device=DEV)
        nat().attention_fwd(q, kv, kv[:, H:], H, 2 * H, 2 * H, mask, ctx, H, lse, B, heads, Sq, Sk, scale, drop, head_dim=d, ctx_f32=o32, causal_tail=tail,
                            keep_bits=kb, keep_lanes=kl)
        return ctx, lse, o32

    # three sites in one launch (different keys), tables pre-filled with garbage
    tabs = [(torch.full((wb,), 0x5A5A5A5A, dtype=torch.int32, device=DEV), torch.full((wl,), -1, dtype=torch.int32, device=DEV)) for _ in drops]
    nat().attention_draw_keep_bits([(dr, B, heads, Sq, Sk, d, kb, kl) for dr, (kb, kl) in zip(drops, tabs)])
    for dr, (kb, kl) in zip(drops, tabs):
        kb_ref = torch.full((wb,), 0x5A5A5A5A, dtype=torch.int32, device=DEV)
        ref = fwd(dr, kb_ref, None)
        assert torch.equal(kb, kb_ref)
        kb_before = kb.clone()
        got = fwd(dr, kb, kl)
        assert torch.equal(kb, kb_before)       # (the reading forward writes no table)
        for a_, b_ in zip(ref, got):
            assert torch.equal(a_, b_)
    assert not torch.equal(tabs[0][0], tabs[1][0])
    # a new seed word: new decisions, still equal to the hashing forward's
    old0 = tabs[0][0].clone()
    seed.add_(1)
    nat().attention_draw_keep_bits([(drops[0], B, heads, Sq, Sk, d) + tabs[0]])
    assert not torch.equal(tabs[0][0], old0)
    ref = fwd(drops[0], None, None); got = fwd(drops[0], None, tabs[0][1])
    for a_, b_ in zip(ref, got):
        assert torch.equal(a_, b_)
    # seed_offset = 1: the decisions of the step AFTER the next advance of the word (what a step draws beside its AdamW launches for its successor)
    nat().attention_draw_keep_bits([(drops[1], B, heads, Sq, Sk, d) + tabs[1]], seed_offset=1)
    seed.add_(1)
    nat().attention_draw_keep_bits([(drops[1], B, heads, Sq, Sk, d) + tabs[2]])
    assert torch.equal(tabs[1][0], tabs[2][0]) and torch.equal(tabs[1][1], tabs[2][1])


def test_attention_draw_refuses_shapes_without_tables():
    drop = nat().drop_cfg(0.1, 1, torch.zeros(1, dtype=torch.int32, device=DEV))
    t = torch.zeros(4096, dtype=torch.int32, device=DEV)
    assert nat().attention_keep_lanes_words(2, 2, 300, 300, 64) == 0
    with pytest.raises(nat().NativeLibraryError):
        nat().attention_draw_keep_bits([(drop, 2, 2, 300, 300, 64, t, t)])
    with pytest.raises(nat().NativeLibraryError):      # dropout off: nothing to draw
        nat().attention_draw_keep_bits([(nat().NO_DROP, 2, 2, 64, 64, 64, t, t)])
    nat().attention_draw_keep_bits([])


def test_attention_keep_bit_table_is_the_counter_hash():
    """Every bit of the table against a host restatement of the dropout RNG (mmf_amd/csrc/common.h: mix24 of (pair index + key), the 16-bit half of the
    element's parity against thr16; element index ((b * heads + head) * Sq + q) * Sk_pad + key): a wrong or stale word cannot hide behind statistics."""
    M32 = 0xFFFFFFFF

    def mix24(x):
        x = x & M32
        x = x ^ (x >> 16)
        x = (((x & 0xFFFFFF) * 0xB5297B) + (((x << 9) | (x >> 23)) & M32)) & M32
        x = x ^ (x >> 13)
        x = (((x & 0xFFFFFF) * 0x68E31F) + (((x << 11) | (x >> 21)) & M32)) & M32
        return x ^ (x >> 15)

    for B, heads, S in ((2, 3, 228), (1, 2, 256), (3, 1, 130), (2, 2, 100), (1, 1, 33)):
        H = heads * 64
        qkv = rnd(B * S, 3 * H, scale=0.5, seed=S + 1)
        drop = nat().drop_cfg(0.1, 424242 + S)
        kb = torch.zeros(nat().attention_keep_bits_words(B, heads, S, S, 64), dtype=torch.int32, device=DEV)
        ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV); lse = torch.empty(B, heads, S, device=DEV)
        nat().attention_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, None, ctx, H, lse, B, heads, S, S, 0.125, drop, keep_bits=kb)
        nt = (S + 31) // 32
        skp = nt * 32
        tab = kb.view(B * heads, nt, nt, 32).long() & M32                                   # [bh, query tile, key tile, key] bit x = query
        bits = torch.stack([(tab >> x) & 1 for x in range(32)], dim=-1)                     # [bh, qt, kt, key, query]
        got = bits.permute(0, 1, 4, 2, 3).reshape(B * heads, skp, skp).bool()[:, :S, :S]    # [bh, q, key]
        bh = torch.arange(B * heads, device=DEV)[:, None, None]; q = torch.arange(S, device=DEV)[None, :, None]; k = torch.arange(S, device=DEV)[None, None, :]
        idx = ((bh * S + q) * skp + k) & M32
        h = mix24((idx >> 1) + drop[0])
        exp = torch.where((k & 1) == 1, h >> 16, h & 0xFFFF) >= drop[1]
        assert torch.equal(got, exp), (B, heads, S, int((got != exp).sum()))


def test_attention_exact_delta_reduces_common_mode_leak():
    """A common component in K (e.g. the key bias) must not reach dQ: sum_key dS = 0.  With delta formed from the bf16 O the
    rows of dS sum to ~2^-9 |dO||O| and that times mean(K) lands in dQ; the fp32 copy of O removes that term (what is
    left is the bf16 rounding of dS itself ahead of the dS K product)."""
    B, heads, S, d = 2, 2, 100, 64
    H = heads * d
    qkv = rnd(B * S, 3 * H, scale=0.3)
    qkv[:, H:2 * H] += 6.0            # large common component in every key
    scale = 1.0 / math.sqrt(d)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    qf = split_heads(q.contiguous(), B, S, heads).requires_grad_(True)
    kf = split_heads(k.contiguous(), B, S, heads).requires_grad_(True)
    vf = split_heads(v.contiguous(), B, S, heads).requires_grad_(True)
    o_ref, _ = attn_ref(qf, kf, vf, None, scale)
    dctx = rnd(B * S, H)
    o_ref.backward(split_heads(dctx, B, S, heads))
    errs = {}
    for exact in (False, True):
        ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV); lse = torch.empty(B, heads, S, device=DEV)
        o32 = torch.empty(B * S, H, device=DEV) if exact else None
        nat().attention_fwd(q, k, v, 3 * H, 3 * H, 3 * H, None, ctx, H, lse, B, heads, S, S, scale, ctx_f32=o32)
        if exact:
            assert torch.equal(o32.bfloat16(), ctx)
        dqkv = torch.zeros_like(qkv); delta = torch.empty(B, heads, S, device=DEV)
        nat().attention_bwd(q, k, v, 3 * H, 3 * H, 3 * H, None, ctx, H, lse, B, heads, S, S, scale, dctx,
                            dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:], delta, ctx_f32=o32)
        g = split_heads(dqkv[:, :H].contiguous(), B, S, heads)
        errs[exact] = float((g - qf.grad).norm() / qf.grad.norm())
    assert errs[True] <= 5e-2, errs
    assert errs[True] < 0.9 * errs[False], errs


def test_attention_fully_masked_rows_are_uniform():
    # additive -10000 (not -inf): a row whose keys are all masked attends uniformly
    B, heads, S = 1, 1, 64
    H = 64
    qkv = rnd(B * S, 3 * H)
    mask = torch.full((B, S), -10000.0, device=DEV)
    ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV); lse = torch.empty(B, heads, S, device=DEV)
    nat().attention_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, heads, S, S, 0.125)
    qf, kf, vf = (split_heads(qkv[:, i * H:(i + 1) * H].contiguous(), B, S, heads) for i in range(3))
    o_ref, _ = attn_ref(qf, kf, vf, mask, 0.125)
    close(split_heads(ctx, B, S, heads), o_ref, 2e-2, 2e-2, "masked rows")
    assert torch.isfinite(ctx.float()).all()


def test_attention_dropout_consistent_between_forward_and_backward():
    # With Sk = 64 and V = I the context IS the dropped probability matrix, which exposes the mask.
    B, heads, S = 2, 2, 64
    H = heads * 64
    qk = rnd(B * S, 2 * H)
    eye = torch.eye(64, device=DEV, dtype=torch.bfloat16)
    v = eye.repeat(B, heads).contiguous()  # [B*S, H]: row s of every head = e_s
    qkv = torch.cat([qk, v], dim=1).contiguous()
    drop = nat().drop_cfg(0.1, 777)
    ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV); lse = torch.empty(B, heads, S, device=DEV)
    scale = 0.125
    nat().attention_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, None, ctx, H, lse, B, heads, S, S, scale, drop)
    qf, kf, vf = (split_heads(qkv[:, i * H:(i + 1) * H].contiguous(), B, S, heads).requires_grad_(True) for i in range(3))
    p_ref = torch.softmax(torch.matmul(qf, kf.transpose(-1, -2)) * scale, dim=-1)
    pd = split_heads(ctx, B, S, heads)  # [B, heads, q, key]
    pmask = (pd > 0.5 * p_ref * drop[2]).float()
    frac = 1.0 - float(pmask.mean())
    assert 0.07 < frac < 0.13, frac
    close(pd, p_ref * pmask * drop[2], 2e-2, 2e-3, "dropped probabilities")
    # backward with the same mask
    dctx = rnd(B * S, H)
    dqkv = torch.zeros_like(qkv); delta = torch.empty(B, heads, S, device=DEV)
    nat().attention_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, None, ctx, H, lse, B, heads, S, S, scale,
                        dctx, dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:], delta, drop)
    o_ref, _ = attn_ref(qf, kf, vf, None, scale, pmask, drop[2])
    o_ref.backward(split_heads(dctx, B, S, heads))
    for name, i, ref_ in (("dq", 0, qf.grad), ("dk", 1, kf.grad), ("dv", 2, vf.grad)):
        g = split_heads(dqkv[:, i * H:(i + 1) * H].contiguous(), B, S, heads)
        close(g, ref_, 3e-2, 3e-2 * float(ref_.abs().max()), name + " (dropout)")


@pytest.mark.parametrize("B,heads,Sq,Sk,tail,p", [(2, 3, 228, 228, 0, 0.1), (1, 2, 37, 37, 0, 0.0), (1, 1, 256, 256, 0, 0.1),
                                                  (2, 2, 100, 228, 0, 0.1), (2, 2, 228, 100, 0, 0.0), (1, 2, 20, 200, 0, 0.1),
                                                  (2, 2, 182, 182, 12, 0.1), (1, 1, 32, 32, 5, 0.0)])
def test_attention_one_pass_backward_agrees_with_the_two_kernel_backward(B, heads, Sq, Sk, tail, p):
    """head_dim 64: the fused backward (one workgroup per (batch, head), dS transposed through LDS, dQ accumulated in LDS) against
    the separate dQ and dK/dV kernels (MMF_TUN_ALT_FORMS bit 2): same dropout decisions, same masks (key mask, prefix-LM tail),
    same gradients up to bf16 rounding of differently ordered sums; rectangular (cross-attention) shapes included."""
    H = heads * 64
    q = rnd(B * Sq, H, scale=1.0); kv = rnd(B * Sk, 2 * H, scale=1.0)
    mbin = (torch.rand(B, Sk, device=DEV) > 0.2).long(); mbin[:, 0] = 1
    mask = torch.empty(B, Sk, device=DEV)
    nat().make_additive_mask(mbin, mask)
    drop = nat().drop_cfg(p, 4242) if p > 0 else nat().NO_DROP
    ctx = torch.empty(B * Sq, H, dtype=torch.bfloat16, device=DEV); o32 = torch.empty(B * Sq, H, device=DEV)
    lse = torch.empty(B, heads, Sq, device=DEV)
    k, v = kv[:, :H], kv[:, H:]
    nat().attention_fwd(q, k, v, H, 2 * H, 2 * H, mask, ctx, H, lse, B, heads, Sq, Sk, 0.125, drop, ctx_f32=o32, causal_tail=tail)
    dctx = rnd(B * Sq, H)
    outs = {}
    try:
        for two_pass in (1, 0):
            nat().set_tunable(nat().TUN_ALT_FORMS, 4 * int(two_pass))
            for exact in (True, False):
                dq = torch.full_like(q, 7.0); dkv = torch.full_like(kv, 7.0); delta = torch.empty(B, heads, Sq, device=DEV)
                nat().attention_bwd(q, k, v, H, 2 * H, 2 * H, mask, ctx, H, lse, B, heads, Sq, Sk, 0.125, dctx, dq, dkv[:, :H], dkv[:, H:],
                                    delta, drop, ctx_f32=o32 if exact else None, causal_tail=tail)
                outs[(two_pass, exact)] = (dq.float(), dkv.float())
    finally:
        nat().set_tunable(nat().TUN_ALT_FORMS, 0)
    for exact in (True, False):
        for name, a_, b_ in (("dq", outs[(0, exact)][0], outs[(1, exact)][0]), ("dk|dv", outs[(0, exact)][1], outs[(1, exact)][1])):
            assert torch.isfinite(a_).all()
            close(a_, b_, 2e-2, 1e-2 * float(b_.abs().max()), "%s one-pass vs two-kernel (exact delta %s)" % (name, exact))
            # no systematic difference: the mean signed deviation is far below one bf16 ulp of the typical magnitude
            assert abs(float((a_ - b_).mean())) <= 1e-3 * float(b_.abs().mean()) + 1e-6, name


def _host_dropout_keep(key, thr16, B, heads, Sq, Sk):
    """The attention kernels' dropout decisions restated on the host (mmf_amd/csrc/common.h mix24 / drop_hash; element index
    ((b * heads + head) * Sq + q) * round_up(Sk, 32) + key, one 32-bit hash per PAIR of elements, 16-bit halves against thr16)."""
    import numpy as np
    M32 = np.uint64(0xFFFFFFFF)
    skp = (Sk + 31) // 32 * 32
    bh = np.arange(B * heads, dtype=np.uint64)[:, None, None]
    q = np.arange(Sq, dtype=np.uint64)[None, :, None]
    k = np.arange(Sk, dtype=np.uint64)[None, None, :]
    e = ((bh * np.uint64(Sq) + q) * np.uint64(skp) + k) & M32
    x = ((e >> np.uint64(1)) + np.uint64(key)) & M32
    rotl = lambda v, r: ((v << np.uint64(r)) | (v >> np.uint64(32 - r))) & M32
    mul24 = lambda v, c: ((v & np.uint64(0xFFFFFF)) * np.uint64(c)) & M32
    x ^= x >> np.uint64(16)
    x = (mul24(x, 0xB5297B) + rotl(x, 9)) & M32
    x ^= x >> np.uint64(13)
    x = (mul24(x, 0x68E31F) + rotl(x, 11)) & M32
    x ^= x >> np.uint64(15)
    half = np.where((e & np.uint64(1)) == 1, x >> np.uint64(16), x & np.uint64(0xFFFF))
    return torch.from_numpy((half >= np.uint64(thr16)).astype(np.float32)).view(B, heads, Sq, Sk).to(DEV)


@pytest.mark.parametrize("B,heads,Sq,Sk,d,tail,p", [
    (2, 2, 320, 320, 64, 0, 0.0), (1, 3, 512, 512, 64, 0, 0.1), (2, 2, 100, 400, 64, 0, 0.1), (1, 2, 300, 120, 64, 0, 0.0), (1, 2, 384, 384, 64, 20, 0.1),
    (1, 1, 1, 300, 64, 0, 0.0), (2, 2, 200, 200, 128, 0, 0.1), (1, 2, 256, 256, 128, 0, 0.0), (1, 2, 60, 250, 128, 0, 0.1), (1, 1, 228, 228, 64, 0, 0.1)])
def test_attention_long_sequences_forward_backward(B, heads, Sq, Sk, d, tail, p):
    """Round 5: any length up to BERT's max_position_embeddings (512) at head_dim 64, up to 256 at head_dim 128 — `BertSelfAttentionJit.forward`
    (mmf/modules/hf_layers.py:161-213) and `BertBiAttention` (mmf/models/vilbert.py:347-475) take any length; round 4 refused more than
    256 / 128.  K and V of a head still sit whole in LDS (128 KB at the caps), the softmax is the same exact two-pass form.  Forward and
    backward against fp32 torch with ragged key masks, the prefix-LM tail, rectangular shapes and dropout (the keep decisions restated on
    the host from the hash); the last case is the VQA2 shape through the same comparison (the tuned kernels)."""
    H = heads * d
    hof = lambda x, S: x.reshape(B, S, heads, d).permute(0, 2, 1, 3).float()
    qbuf = rnd(B * Sq, 3 * H, seed=31); kvbuf = rnd(B * Sk, 3 * H, seed=32) if (Sq != Sk or tail == 0) else None
    if tail:
        kvbuf = qbuf
    q, k, v = qbuf[:, :H], kvbuf[:, H:2 * H], kvbuf[:, 2 * H:]
    mbin = (torch.rand(B, Sk, device=DEV) > 0.2).long(); mbin[:, 0] = 1
    if tail:
        mbin[:, Sk - tail:] = 1
    mask = torch.empty(B, Sk, device=DEV); nat().make_additive_mask(mbin, mask)
    scale = 1.0 / math.sqrt(d)
    drop = nat().drop_cfg(p, 99173) if p > 0 else nat().NO_DROP
    ctx = torch.full((B * Sq, H), float("nan"), dtype=torch.bfloat16, device=DEV); lse = torch.full((B, heads, Sq), float("nan"), device=DEV)
    o32 = torch.full((B * Sq, H), float("nan"), device=DEV)
    nat().attention_fwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, heads, Sq, Sk, scale, drop, head_dim=d, ctx_f32=o32, causal_tail=tail)
    qf = hof(q.contiguous(), Sq).requires_grad_(True); kf = hof(k.contiguous(), Sk).requires_grad_(True); vf = hof(v.contiguous(), Sk).requires_grad_(True)
    add = mask[:, None, None, :].expand(B, 1, Sq, Sk).clone()
    if tail:      # m4c.py:424-440: a decoding key (the last `tail` positions) is visible to the decoding queries at or after it only, whatever the key mask says
        cf = Sk - tail
        qi = torch.arange(Sq, device=DEV)[:, None]; ki = torch.arange(Sk, device=DEV)[None, :]
        causal = torch.where((qi >= cf) & (ki <= qi), 0.0, -10000.0).to(DEV)
        add = torch.where((ki >= cf)[None, None], causal[None, None], add)
    sc = torch.matmul(qf, kf.transpose(-1, -2)) * scale + add
    pr = torch.softmax(sc, dim=-1)
    lse_ref = torch.logsumexp(sc, dim=-1)
    if p > 0:
        pr = pr * _host_dropout_keep(drop[0], drop[1], B, heads, Sq, Sk) * drop[2]
    o_ref = torch.matmul(pr, vf)
    assert torch.isfinite(ctx.float()).all() and torch.isfinite(lse).all()
    close(hof(ctx, Sq), o_ref, 2e-2, 2e-2, "ctx")
    close(lse, lse_ref, 1e-4, 3e-3, "lse")
    assert torch.equal(o32.bfloat16(), ctx)
    dctx = rnd(B * Sq, H, seed=33)
    dqb = torch.zeros_like(qbuf); dkvb = torch.zeros_like(kvbuf) if kvbuf is not qbuf else dqb
    delta = torch.empty(B, heads, Sq, device=DEV)
    nat().attention_bwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, heads, Sq, Sk, scale, dctx,
                        dqb[:, :H], dkvb[:, H:2 * H], dkvb[:, 2 * H:], delta, drop, head_dim=d, ctx_f32=o32, causal_tail=tail)
    o_ref.backward(hof(dctx, Sq))
    for name, got, ref, S in (("dq", dqb[:, :H], qf.grad, Sq), ("dk", dkvb[:, H:2 * H], kf.grad, Sk), ("dv", dkvb[:, 2 * H:], vf.grad, Sk)):
        close(hof(got.contiguous(), S), ref, 3e-2, 3e-2 * float(ref.abs().max()), name)


def test_attention_length_caps_are_reported():
    from mmf_amd._native import NativeLibraryError
    H = 64
    qkv = rnd(520, 3 * H)
    ctx = torch.empty(520, H, dtype=torch.bfloat16, device=DEV); lse = torch.empty(1, 1, 520, device=DEV)
    with pytest.raises(NativeLibraryError, match="<= 512"):
        nat().attention_fwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, None, ctx, H, lse, 1, 1, 520, 520, 0.125)
    q2 = rnd(300, 3 * 128)
    with pytest.raises(NativeLibraryError, match="head_dim 128 is built for Sq, Sk <= 256"):
        nat().attention_fwd(q2, q2[:, 128:], q2[:, 256:], 384, 384, 384, None, torch.empty(300, 128, dtype=torch.bfloat16, device=DEV), 128,
                            torch.empty(1, 1, 300, device=DEV), 1, 1, 300, 300, 0.1, head_dim=128)
    # (a per-query mask runs at every length the key mask does since round 6: test_attention_per_query_mask_against_reference)
    nat().attention_fwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, torch.zeros(1, 300, 300, device=DEV), ctx, H, lse, 1, 1, 300, 300, 0.125)


# ---------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,H", [(7296, 768), (33, 768), (64, 1024), (5, 64), (12, 256), (20011, 768), (100, 512), (14592, 1024)])
def test_layernorm_forward_backward(rows, H):
    x = rnd(rows, H, scale=2.0); gamma = rnd(H, dtype=torch.float32) + 1.0; beta = rnd(H, dtype=torch.float32)
    y = torch.empty_like(x); mean = torch.empty(rows, device=DEV); rstd = torch.empty(rows, device=DEV)
    nat().layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, H, 1e-12)
    xf = x.float().requires_grad_(True); g = gamma.clone().requires_grad_(True); b = beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xf, (H,), g, b, 1e-12)
    close(y, ref, 1e-2, 1e-2, "ln fwd")
    close(mean, xf.mean(-1), 1e-4, 1e-5, "mean")
    dy = rnd(rows, H)
    dx = torch.empty_like(x)
    dgamma = torch.empty(H, device=DEV); dbeta = torch.empty(H, device=DEV); dbias = torch.empty(H, device=DEV)
    ws = torch.empty(nat().layernorm_bwd_ws_floats(H), device=DEV)
    nat().layernorm_bwd(dy, x, mean, rstd, gamma, dx, None, (0, 0, 1.0), dgamma, dbeta, dbias, 0, ws, rows, H)
    ref.backward(dy.float())
    close(dx, xf.grad, 1e-2, 1e-2 * float(xf.grad.abs().max()), "ln dx")
    close(dgamma, g.grad, 1e-3, 1e-3 * float(g.grad.abs().max()) + 1e-4, "dgamma")
    close(dbeta, b.grad, 1e-3, 1e-3 * float(b.grad.abs().max()) + 1e-4, "dbeta")
    close(dbias, dx.float().sum(0), 1e-3, 1e-3 * float(dx.float().sum(0).abs().max()) + 1e-4, "dbias")
    # accumulate flag
    nat().layernorm_bwd(dy, x, mean, rstd, gamma, dx, None, (0, 0, 1.0), dgamma, dbeta, None, 1, ws, rows, H)
    close(dgamma, 2 * g.grad, 1e-3, 2e-3 * float(g.grad.abs().max()) + 1e-4, "dgamma accumulate")


@pytest.mark.parametrize("H", [768, 1024])
def test_layernorm_half_wave_kernels_agree_with_the_one_wave_per_row_kernels(H):
    """H % 256 == 0 runs the half-wave-per-row, 16-byte kernels; MMF_TUN_ALT_FORMS bit 0 selects the one-wave-per-row form.  Same maths:
    outputs equal up to fp32 summation order (bf16 outputs almost always identical), dropout masks identical."""
    rows = 3000
    x = rnd(rows, H, scale=2.0, seed=1); gamma = rnd(H, dtype=torch.float32, seed=2) + 1.0; beta = rnd(H, dtype=torch.float32, seed=3)
    dy = rnd(rows, H, seed=4)
    drop = nat().drop_cfg(0.1, 77)
    res = []
    for old in (0, 1):
        nat().set_tunable(nat().TUN_ALT_FORMS, old)
        try:
            y = torch.empty_like(x); mean = torch.empty(rows, device=DEV); rstd = torch.empty(rows, device=DEV)
            nat().layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, H, 1e-12)
            dx = torch.empty_like(x); dlin = torch.empty_like(x)
            dg = torch.empty(H, device=DEV); db = torch.empty(H, device=DEV)
            ws = torch.empty(nat().layernorm_bwd_ws_floats(H), device=DEV)
            nat().layernorm_bwd(dy, x, mean, rstd, gamma, dx, dlin, drop, dg, db, None, 0, ws, rows, H)
            res.append((y, mean, rstd, dx, dlin, dg, db))
        finally:
            nat().set_tunable(nat().TUN_ALT_FORMS, 0)
    for a, b, name in zip(res[0], res[1], ("y", "mean", "rstd", "dx", "dlin", "dgamma", "dbeta")):
        close(a, b, 1e-2 if a.dtype == torch.bfloat16 else 1e-5, 1e-2 if a.dtype == torch.bfloat16 else 1e-4 * float(b.abs().max()), name)
    assert torch.equal(res[0][4] == 0, res[1][4] == 0)      # same dropout mask


def _attn_run(qkv, mask, B, heads, S, drop, tail=0):
    H = heads * 64
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    ctx = torch.full((B * S, H), float("nan"), dtype=torch.bfloat16, device=DEV); c32 = torch.full((B * S, H), float("nan"), device=DEV)
    lse = torch.full((B, heads, S), float("nan"), device=DEV)
    nat().attention_fwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, heads, S, S, 0.125, drop, ctx_f32=c32, causal_tail=tail)
    dctx = rnd(B * S, H, seed=77)
    dqkv = torch.full_like(qkv, 5.0); delta = torch.empty(B, heads, S, device=DEV)
    nat().attention_bwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, heads, S, S, 0.125, dctx, dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:],
                        delta, drop, ctx_f32=c32, causal_tail=tail)
    return ctx, c32, lse, dqkv


@pytest.mark.parametrize("B,heads,S", [(2, 3, 228), (2, 2, 100), (1, 2, 182), (1, 1, 33), (1, 2, 256), (2, 2, 300), (1, 2, 384), (1, 1, 512)])
def test_attention_per_query_mask_against_reference(B, heads, S):
    """A materialised additive mask per (query, key) pair, [B, S, S] — `attention_scores + attention_mask` with a [B, 1, S, S] mask
    (mmf/modules/hf_layers.py:187-190) — read by the forward kernels (every form) and the backward (one-pass up to 256 positions, the dQ and dK / dV
    kernels beyond: BERT's 512 max_position_embeddings): against fp32 torch."""
    H = heads * 64
    qkv = rnd(B * S, 3 * H, scale=1.0, seed=31)
    g = torch.Generator(device="cpu").manual_seed(S)
    vis = torch.rand(B, S, S, generator=g) > 0.3
    vis[:, torch.arange(S), torch.arange(S)] = True          # every query sees itself
    mask3 = ((~vis).float() * -10000.0).to(DEV)
    # mixed-magnitude entries too: the mask is ADDED, not a switch
    mask3 = mask3 + (torch.rand(B, S, S, generator=g) * 2.0 - 1.0).to(DEV)
    ctx, c32, lse, dqkv = _attn_run(qkv, mask3, B, heads, S, nat().NO_DROP)
    qf = split_heads(qkv[:, :H].contiguous(), B, S, heads).requires_grad_(True)
    kf = split_heads(qkv[:, H:2 * H].contiguous(), B, S, heads).requires_grad_(True)
    vf = split_heads(qkv[:, 2 * H:].contiguous(), B, S, heads).requires_grad_(True)
    s_ = torch.matmul(qf, kf.transpose(-1, -2)) * 0.125 + mask3[:, None]
    o_ref = torch.matmul(torch.softmax(s_, dim=-1), vf)
    close(split_heads(ctx, B, S, heads), o_ref, 2e-2, 2e-2, "ctx")
    close(lse, torch.logsumexp(s_, dim=-1), 1e-4, 2e-3, "lse")
    assert torch.equal(c32.bfloat16(), ctx)
    o_ref.backward(split_heads(rnd(B * S, H, seed=77), B, S, heads))
    for name, got_, ref_ in (("dq", dqkv[:, :H], qf.grad), ("dk", dqkv[:, H:2 * H], kf.grad), ("dv", dqkv[:, 2 * H:], vf.grad)):
        close(split_heads(got_.contiguous(), B, S, heads), ref_, 3e-2, 3e-2 * float(ref_.abs().max()), name)


@pytest.mark.parametrize("B,heads,S,tail", [(2, 3, 228, 0), (2, 2, 96, 0), (2, 2, 182, 12), (1, 2, 64, 7), (2, 2, 300, 0), (1, 2, 420, 30)])
def test_attention_per_query_mask_is_bit_identical_to_the_structured_forms(B, heads, S, tail):
    """The same mask handed over as the key mask [B, S] (+ M4C's causal tail, mmf/models/m4c.py:424-440) and materialised as [B, S, S]: context,
    its fp32 copy, log-sum-exp and dQ | dK | dV are bit-identical, dropout included (same decisions: they depend on the element index only)."""
    H = heads * 64
    qkv = rnd(B * S, 3 * H, scale=1.0, seed=41)
    key = torch.zeros(B, S, device=DEV)
    for b in range(B):
        key[b, S - tail - 3 - 5 * b:S - tail] = -10000.0
    mask3 = key[:, None, :].expand(B, S, S).clone()
    if tail:
        q = torch.arange(S, device=DEV)[:, None]; k = torch.arange(S, device=DEV)[None, :]
        c0 = S - tail
        dec = (k >= c0)
        mask3[:, dec & ~((q >= c0) & (k <= q))] = -10000.0
        mask3[:, dec & (q >= c0) & (k <= q)] = 0.0
    drop = nat().drop_cfg(0.1, 4321)
    a = _attn_run(qkv, key, B, heads, S, drop, tail)
    b_ = _attn_run(qkv, mask3.contiguous(), B, heads, S, drop, 0)
    for name, x, y in zip(("ctx", "ctx32", "lse", "dqkv"), a, b_):
        assert torch.isfinite(x.float()).all(), name
        assert torch.equal(x, y), name


@pytest.mark.parametrize("B,heads,S", [(2, 2, 228), (1, 2, 100)])
def test_attention_per_query_mask_two_kernel_backward_matches_the_one_pass_kernel(B, heads, S):
    """The dQ and dK / dV kernels read a per-query mask too (what runs beyond 256 positions); forced at a one-pass shape (MMF_TUN_ALT_FORMS bit 2)
    their gradients equal the one-pass kernel's up to the bf16 rounding of differently ordered sums."""
    H = heads * 64
    qkv = rnd(B * S, 3 * H, scale=1.0, seed=51)
    g = torch.Generator(device="cpu").manual_seed(S)
    mask3 = ((torch.rand(B, S, S, generator=g) < 0.3).float() * -10000.0).to(DEV)
    drop = nat().drop_cfg(0.1, 99)
    outs = []
    try:
        for two in (False, True):
            nat().set_tunable(nat().TUN_ALT_FORMS, 4 * int(two))
            outs.append(_attn_run(qkv, mask3, B, heads, S, drop))
    finally:
        nat().set_tunable(nat().TUN_ALT_FORMS, 0)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2])
    close(outs[1][3], outs[0][3], 2e-2, 2e-2 * float(outs[0][3].abs().max()), "dqkv")


@pytest.mark.parametrize("B,heads,S", [(2, 3, 228), (2, 2, 100), (1, 2, 300), (1, 4, 33)])
def test_attention_per_head_mask_against_reference(B, heads, S):
    """One additive [S, S] mask per (sample, head), [B, heads, S, S] — what `attention_scores + attention_mask` (mmf/modules/hf_layers.py:187-190)
    does with a mask that is not broadcast over the heads — read through mmf_attn_desc.mask_head_stride by every forward and backward form: against
    fp32 torch; and a mask repeated over the heads gives the bits of the [B, S, S] form."""
    H = heads * 64
    qkv = rnd(B * S, 3 * H, scale=1.0, seed=61)
    g = torch.Generator(device="cpu").manual_seed(S + heads)
    vis = torch.rand(B, heads, S, S, generator=g) > 0.3
    vis[:, :, torch.arange(S), torch.arange(S)] = True
    mask4 = (((~vis).float() * -10000.0) + torch.rand(B, heads, S, S, generator=g) * 2.0 - 1.0).to(DEV).contiguous()
    ctx, c32, lse, dqkv = _attn_run(qkv, mask4, B, heads, S, nat().NO_DROP)
    qf = split_heads(qkv[:, :H].contiguous(), B, S, heads).requires_grad_(True)
    kf = split_heads(qkv[:, H:2 * H].contiguous(), B, S, heads).requires_grad_(True)
    vf = split_heads(qkv[:, 2 * H:].contiguous(), B, S, heads).requires_grad_(True)
    s_ = torch.matmul(qf, kf.transpose(-1, -2)) * 0.125 + mask4
    o_ref = torch.matmul(torch.softmax(s_, dim=-1), vf)
    close(split_heads(ctx, B, S, heads), o_ref, 2e-2, 2e-2, "ctx")
    close(lse, torch.logsumexp(s_, dim=-1), 1e-4, 2e-3, "lse")
    o_ref.backward(split_heads(rnd(B * S, H, seed=77), B, S, heads))
    for name, got_, ref_ in (("dq", dqkv[:, :H], qf.grad), ("dk", dqkv[:, H:2 * H], kf.grad), ("dv", dqkv[:, 2 * H:], vf.grad)):
        close(split_heads(got_.contiguous(), B, S, heads), ref_, 3e-2, 3e-2 * float(ref_.abs().max()), name)
    drop = nat().drop_cfg(0.1, 777)
    a = _attn_run(qkv, mask4[:, 0].contiguous(), B, heads, S, drop)
    b_ = _attn_run(qkv, mask4[:, :1].expand(B, heads, S, S).contiguous(), B, heads, S, drop)
    for name, x, y in zip(("ctx", "ctx32", "lse", "dqkv"), a, b_):
        assert torch.equal(x, y), name


def test_attention_per_query_mask_is_refused_where_it_is_not_built():
    from mmf_amd._native import NativeLibraryError
    B, heads, S, H = 1, 1, 64, 128
    qkv = rnd(B * S, 3 * H)
    ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV); lse = torch.empty(B, heads, S, device=DEV)
    m3 = torch.zeros(B, S, S, device=DEV)
    with pytest.raises(NativeLibraryError, match="head_dim 64"):
        nat().attention_fwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, m3, ctx, H, lse, B, heads, S, S, 0.1, head_dim=128)
    with pytest.raises(NativeLibraryError, match="causal tail"):
        nat().attention_fwd(qkv, qkv[:, 64:], qkv[:, 128:], 3 * H, 3 * H, 3 * H, m3, ctx, 64, lse, B, 1, S, S, 0.1, causal_tail=4)
    with pytest.raises(NativeLibraryError, match=r"\[B, Sq, Sk\]"):
        nat().attention_fwd(qkv, qkv[:, 64:], qkv[:, 128:], 3 * H, 3 * H, 3 * H, torch.zeros(B, S, S + 8, device=DEV), ctx, 64, lse, B, 1, S, S, 0.1)


# ---------------------------------------------------------------------------------------------
# embeddings / row utilities / loss / optimizer
# ---------------------------------------------------------------------------------------------
def test_layernorm_backward_deferred_column_sums_are_bit_identical():
    """dgamma / dbeta of several LayerNorm backwards finished by ONE mmf_layernorm_bwd_reduce_multi launch (what the graphed
    training steps do) against the per-call reduction: same partials, same summation order."""
    cases = [(7296, 768), (100, 256), (3200, 1024), (37, 512)]
    direct, items, keep = [], [], []
    for rows, H in cases:
        assert nat().layernorm_bwd_deferrable(rows, H)
        dy = rnd(rows, H); x = rnd(rows, H, scale=1.0)
        mean = torch.randn(rows, device=DEV) * 0.1; rstd = torch.rand(rows, device=DEV) + 0.5
        gamma = torch.rand(H, device=DEV) + 0.5
        outs = []
        for deferred in (False, True):
            dx = torch.empty_like(dy)
            dg = torch.full((H,), 7.0, device=DEV); db = torch.full((H,), 7.0, device=DEV)
            ws = torch.empty(nat().layernorm_bwd_ws_floats(H), device=DEV)
            if deferred:
                nat().layernorm_bwd(dy, x, mean, rstd, gamma, dx, None, nat().NO_DROP, None, None, None, 0, ws, rows, H)
                items.append((ws, rows, H, dg, db))
            else:
                nat().layernorm_bwd(dy, x, mean, rstd, gamma, dx, None, nat().NO_DROP, dg, db, None, 0, ws, rows, H)
            outs.append((dx, dg, db))
        direct.append(outs)
    nat().layernorm_bwd_reduce_multi(items)
    for (a, b), (rows, H) in zip(direct, cases):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), (rows, H)
    assert not nat().layernorm_bwd_deferrable(100, 300)          # (shapes outside the half-wave kernels are reduced per call)


def test_embed_text_and_scatter_add():
    B, T, S, H, V = 4, 16, 24, 768, 1000
    ids = torch.randint(0, V, (B, T), device=DEV); seg = torch.randint(0, 2, (B, T), device=DEV)
    word = rnd(V, H, dtype=torch.float32); pos = rnd(64, H, dtype=torch.float32); typ = rnd(2, H, dtype=torch.float32)
    y = torch.zeros(B * S, H, dtype=torch.bfloat16, device=DEV)
    nat().embed_text_fwd(ids, seg, word, pos, typ, y, B, T, S, H)
    ref = word[ids] + pos[torch.arange(T, device=DEV)][None] + typ[seg]
    close(y.view(B, S, H)[:, :T], ref, 1e-2, 1e-2, "embed text")
    # backward scatters
    d = rnd(B * S, H)
    d3 = d.view(B, S, H).float()
    dword = torch.zeros(V, H, device=DEV)
    nat().rows_scatter_add(d, H, B, T, S, ids, T, 0, 0, dword, H, 0)
    close(dword, torch.zeros(V, H, device=DEV).index_add_(0, ids.view(-1), d3[:, :T].reshape(-1, H)), 1e-4, 1e-4, "dword")
    dpos = torch.zeros(64, H, device=DEV)
    nat().rows_scatter_add(d, H, B, T, S, None, 0, 1, 0, dpos, H, 0)
    close(dpos[:T], d3[:, :T].sum(0), 1e-4, 1e-4, "dpos")
    dtyp = torch.zeros(2, H, device=DEV)
    nat().rows_scatter_add(d, H, B, T, S, seg, T, 0, 0, dtyp, H, 1)
    close(dtyp, torch.zeros(2, H, device=DEV).index_add_(0, seg.view(-1), d3[:, :T].reshape(-1, H)), 1e-4, 1e-3, "dtype")
    # visual rows: bucket 0 for every row (position_ids_visual = 0, embeddings.py:411-419)
    dpv = torch.zeros(4, H, device=DEV)
    nat().rows_scatter_add(d[T:], H, B, S - T, S, None, 0, 0, 0, dpv, H, 1)
    close(dpv[0], d3[:, T:].sum((0, 1)), 1e-4, 1e-3, "dpos_visual")


@pytest.mark.parametrize("B,T,R,H", [(32, 128, 100, 768), (3, 7, 0, 256), (5, 16, 9, 1024), (2, 4, 3, 100)])
def test_embed_tables_bwd_is_the_four_scatter_adds(B, T, R, H):
    """mmf_embed_tables_bwd (text positions, text token types, visual token types, the visual position row in two launches) against torch and against
    the per-table mmf_rows_scatter_add launches it replaces; outputs are added to (the aligned-position term lands in dpos afterwards)."""
    S, P, NT, NTV, PV = T + R, T + 5, 2, 2, 1
    d = rnd(B * S, H, seed=B * 1000 + H)
    d3 = d.view(B, S, H).float()
    seg = torch.randint(0, NT, (B, T), device=DEV)
    vt = torch.randint(0, NTV, (B, R), device=DEV) if R else None
    base = rnd(P + NT + NTV + PV, H, dtype=torch.float32, seed=7)
    got = base.clone()
    dpos, dtyp, dtv, dpv = got[:P], got[P:P + NT], got[P + NT:P + NT + NTV], got[P + NT + NTV:]
    nat().embed_tables_bwd(d, H, B, T, R, seg, vt, 2, dpos, dtyp, dtv if R else None, dpv if R else None, H)
    ref = base.clone().double()
    ref[2:2 + T] += d3[:, :T].double().sum(0)
    ref[P:P + NT].index_add_(0, seg.view(-1), d3[:, :T].reshape(-1, H).double())
    if R:
        ref[P + NT:P + NT + NTV].index_add_(0, vt.view(-1), d3[:, T:].reshape(-1, H).double())
        ref[P + NT + NTV] += d3[:, T:].double().sum((0, 1))
    close(got, ref.float(), 1e-5, 1e-4 * math.sqrt(B * max(T, R, 1)), "embed_tables_bwd")
    if H % 4 == 0:
        old = torch.zeros_like(base)
        nat().rows_scatter_add(d, H, B, T, S, None, 0, 1, 2, old[:P], H, 0)
        nat().rows_scatter_add(d, H, B, T, S, seg, T, 0, 0, old[P:P + NT], H, 1)
        if R:
            nat().rows_scatter_add(d[T:], H, B, R, S, vt, R, 0, 0, old[P + NT:P + NT + NTV], H, 1)
            nat().rows_scatter_add(d[T:], H, B, R, S, None, 0, 0, 0, old[P + NT + NTV:], H, 1)
        close(got - base, old, 1e-5, 1e-4 * math.sqrt(B * max(T, R, 1)), "embed_tables_bwd vs rows_scatter_add")
    assert nat().take_index_error() is False
    # a token type outside its table is skipped and flagged; the position sums do not depend on it
    bad = seg.clone(); bad[0, 0] = NT + 3
    got2 = torch.zeros_like(base)
    nat().embed_tables_bwd(d, H, B, T, R, bad, vt, 2, got2[:P], got2[P:P + NT], got2[P + NT:P + NT + NTV] if R else None, got2[P + NT + NTV:] if R else None, H)
    assert nat().take_index_error() is True
    close(got2[:P], (got - base)[:P], 1e-5, 1e-5, "positions unaffected by a bad type id")      # (`got` carries `base`: one fp32 rounding apart)
    keep = torch.ones(B, T, dtype=torch.bool, device=DEV); keep[0, 0] = False
    ok = torch.zeros(NT, H, device=DEV, dtype=torch.float64).index_add_(0, seg[keep], d3[:, :T][keep].double())
    close(got2[P:P + NT], ok.float(), 1e-5, 1e-4 * math.sqrt(B * T), "in-range rows still accumulate")


@pytest.mark.parametrize("rows,H", [(7296, 768), (100, 256), (33, 1024)])
def test_layernorm_with_fused_dropout_is_bit_identical_to_two_launches(rows, H):
    """dropout(LayerNorm(x)) of BertVisioLinguisticEmbeddings (embeddings.py:343-345) as one launch each way == mmf_layernorm_fwd + mmf_dropout_bf16 and
    mmf_dropout_bf16 + mmf_layernorm_bwd, bit for bit (deferred and immediate column sums)."""
    assert nat().layernorm_dropout_fusable(H) and not nat().layernorm_dropout_fusable(300)
    x = rnd(rows, H, seed=rows + H); gamma = rnd(H, dtype=torch.float32, seed=1) + 1.0; beta = rnd(H, dtype=torch.float32, seed=2)
    drop = nat().drop_cfg(0.1, 1234567)
    y0 = torch.empty_like(x); y1 = torch.empty_like(x); y = torch.empty_like(x)
    mean0 = torch.empty(rows, device=DEV); rstd0 = torch.empty(rows, device=DEV); mean1 = torch.empty(rows, device=DEV); rstd1 = torch.empty(rows, device=DEV)
    nat().layernorm_fwd(x, gamma, beta, y0, mean0, rstd0, rows, H, 1e-12)
    nat().dropout(y0, y1, drop)
    nat().layernorm_dropout_fwd(x, gamma, beta, y, mean1, rstd1, rows, H, 1e-12, drop)
    assert torch.equal(y, y1) and torch.equal(mean0, mean1) and torch.equal(rstd0, rstd1)
    assert 0.05 < float((y == 0).float().mean()) < 0.15
    dy = rnd(rows, H, seed=5)
    d2 = torch.empty_like(dy)
    nat().dropout(dy, d2, drop)
    ws = torch.empty(nat().layernorm_bwd_ws_floats(H), device=DEV)
    dx0 = torch.empty_like(x); dg0 = torch.empty(H, device=DEV); db0 = torch.empty(H, device=DEV)
    # (the fused form keeps ONE row in flight per half-wave — with two, the extra hash registers would cost the H = 768 kernel its second wave per SIMD —
    # so the two-launch reference runs the same row schedule: MMF_TUN_ALT_FORMS bit 3)
    nat().set_tunable(nat().TUN_ALT_FORMS, 8)
    try:
        nat().layernorm_bwd(d2, x, mean0, rstd0, gamma, dx0, None, nat().NO_DROP, dg0, db0, None, 0, ws, rows, H)
    finally:
        nat().set_tunable(nat().TUN_ALT_FORMS, 0)
    dxn = torch.empty_like(x); dgn = torch.empty(H, device=DEV); dbn = torch.empty(H, device=DEV)
    nat().layernorm_bwd(d2, x, mean0, rstd0, gamma, dxn, None, nat().NO_DROP, dgn, dbn, None, 0, ws, rows, H)      # the default (two-row) schedule
    close(dxn, dx0, 1e-2, 1e-3, "two-row vs one-row schedule dx"); close(dgn, dg0, 1e-5, 1e-3, "dgamma"); close(dbn, db0, 1e-5, 1e-3, "dbeta")
    dx1 = torch.empty_like(x); dg1 = torch.empty(H, device=DEV); db1 = torch.empty(H, device=DEV)
    nat().layernorm_bwd_din(dy, x, mean0, rstd0, gamma, dx1, drop, dg1, db1, 0, ws, rows, H)
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    if nat().layernorm_bwd_deferrable(rows, H):
        dx2 = torch.empty_like(x); dg2 = torch.empty(H, device=DEV); db2 = torch.empty(H, device=DEV)
        nat().layernorm_bwd_din(dy, x, mean0, rstd0, gamma, dx2, drop, None, None, 0, ws, rows, H)
        nat().layernorm_bwd_reduce_multi([(ws, rows, H, dg2, db2)])
        assert torch.equal(dx0, dx2) and torch.equal(dg0, dg2) and torch.equal(db0, db2)
    with pytest.raises(nat().NativeLibraryError):
        nat().layernorm_dropout_fwd(x[:, :300].contiguous(), gamma[:300].contiguous(), beta[:300].contiguous(), y[:, :300].contiguous(), mean1, rstd1, rows, 300, 1e-12, drop)


def test_step_advance_is_seed_advance_plus_optim_state_advance():
    seed = torch.tensor([41], dtype=torch.int32, device=DEV); state = torch.tensor([9.0, 0.0], device=DEV)
    seed2 = seed.clone(); state2 = state.clone()
    nat().seed_advance(seed2); nat().optim_state_advance(state2, 1, 4.0, 100.0)
    nat().step_advance(seed, state, 1, 4.0, 100.0)
    assert torch.equal(seed, seed2) and torch.equal(state, state2) and int(seed) == 42 and float(state[0]) == 10.0
    nat().step_advance(None, state, 0)
    nat().step_advance(seed, None)
    assert int(seed) == 43 and state.tolist() == [11.0, 1.0]


@pytest.mark.parametrize("B,T,S,H,V,dups", [(32, 128, 228, 768, 30522, 0.0), (8, 64, 64, 768, 50, 0.0), (3, 40, 57, 1024, 1000, 0.9), (2, 30, 30, 1280, 7, 0.0),
                                            (12, 128, 128, 768, 2000, 0.8), (2, 60, 60, 768, 500, 0.3)])
def test_rows_scatter_add_with_an_index_array_is_deterministic_and_atomic_free(B, T, S, H, V, dups):
    """mmf_rows_scatter_add with an index array (the word-embedding gradient, embeddings.py:329-345 backward): one owner workgroup per distinct id adds that
    id's rows in row order and writes the output row once — against float64 torch, bit-identical from run to run and in row order (a float32 running sum in
    that order reproduces it exactly), and against the fp32-atomic kernel it replaces; the padding id is skipped, the output is added to.
    Round 6: an id on more than 64 rows (M4C's previous-prediction gather, m4c.py:284-304: most of 1536 rows carry index 0; the (12, 128, ...) case) is
    added by all of its rows with fp32 atomics instead of serially by one owner: right to rounding, every OTHER row still bit-reproducible."""
    g = torch.Generator(device="cpu"); g.manual_seed(B * T + H)
    ids = torch.randint(0, V, (B, T), generator=g)
    if dups:
        ids[torch.rand(B, T, generator=g) < dups] = 3          # one id for most rows: a long run for a single owner
    ids[0, :2] = 0                                             # some padding tokens
    ids = ids.to(DEV)
    d = rnd(B * S, H, seed=V)
    base = rnd(V, H, dtype=torch.float32, seed=1)
    outs = []
    for rep in range(3):
        out = base.clone()
        nat().rows_scatter_add(d, H, B, T, S, ids, T, 0, 0, out, H, 0, 0)       # skip_bucket = 0: nn.Embedding(padding_idx=0)
        outs.append(out)
    rows = d.view(B, S, H)[:, :T].reshape(-1, H).float()
    flat = ids.view(-1)
    keep = flat != 0
    light = torch.bincount(flat[keep], minlength=V) <= 64      # (SCATTER_RUN_MAX: longer runs take the atomic path)
    assert bool(light.all()) == (dups * B * T < 64), "the parametrisation covers both sides of SCATTER_RUN_MAX"
    assert torch.equal(outs[0][light], outs[1][light]) and torch.equal(outs[0][light], outs[2][light])
    ref = base.double().index_add_(0, flat[keep], rows[keep].double())
    close(outs[0], ref.float(), 1e-5, 1e-4 * math.sqrt(B * T), "owner-wave scatter add")
    assert torch.equal(outs[0][0], base[0])                    # the padding row received nothing
    # exact order: float32 running sums in row order
    for bucket in [b_ for b_ in flat[keep].unique().tolist() if bool(light[b_])][:5]:
        acc = torch.zeros(H, device=DEV)
        for r in torch.nonzero(flat == bucket).view(-1).tolist():
            acc = acc + rows[r]
        assert torch.equal(outs[0][bucket], base[bucket] + acc), bucket
    nat().set_tunable(nat().TUN_SCATTER_ATOMIC, 1)             # MMF_TUN_SCATTER_ATOMIC: the fp32-atomic kernel
    try:
        old = base.clone()
        nat().rows_scatter_add(d, H, B, T, S, ids, T, 0, 0, old, H, 0, 0)
    finally:
        nat().set_tunable(nat().TUN_SCATTER_ATOMIC, 0)
    close(outs[0], old, 1e-5, 1e-4 * math.sqrt(B * T), "owner-wave vs atomics")


def test_embedding_indices_are_bounded_like_nn_embedding():
    """nn.Embedding raises IndexError for an id outside its table; here the host cannot see device ids without a sync, so the
    kernels skip the offending rows (forward: zero row; backward: no atomic write out of bounds) and raise a device flag that
    `take_index_error()` reports.  A sequence longer than the position table is refused on the host."""
    B, T, S, H, V = 2, 8, 8, 256, 50
    word = rnd(V, H, dtype=torch.float32); pos = rnd(16, H, dtype=torch.float32); typ = rnd(2, H, dtype=torch.float32)
    ids = torch.randint(0, V, (B, T), device=DEV); seg = torch.zeros(B, T, dtype=torch.int64, device=DEV)
    y = torch.full((B * S, H), 7.0, dtype=torch.bfloat16, device=DEV)
    assert nat().take_index_error() is False
    nat().embed_text_fwd(ids, seg, word, pos, typ, y, B, T, S, H)
    assert nat().take_index_error() is False
    bad = ids.clone(); bad[1, 3] = V + 5; bad[0, 0] = -1
    seg2 = seg.clone(); seg2[1, 6] = 2
    nat().embed_text_fwd(bad, seg2, word, pos, typ, y, B, T, S, H)
    assert nat().take_index_error() is True and nat().take_index_error() is False        # reported once, then cleared
    yv = y.view(B, S, H).float()
    assert float(yv[1, 3].abs().max()) == 0.0 and float(yv[0, 0].abs().max()) == 0.0 and float(yv[1, 6].abs().max()) == 0.0
    close(yv[0, 1], word[ids[0, 1]] + pos[1] + typ[0], 1e-2, 1e-2, "in-range rows unaffected")
    with pytest.raises(nat().NativeLibraryError):          # 8 positions starting at 12 overrun the 16-row position table
        nat().embed_text_fwd(ids, seg, word, pos, typ, y, B, T, S, H, 0, 12)
    # backward: an index past the table must not be written anywhere
    d = rnd(B * S, H)
    guard = torch.zeros(V + 64, H, device=DEV)             # rows [V, V + 64) play the memory behind the table
    nat().rows_scatter_add(d, H, B, T, S, bad, T, 0, 0, guard[:V], H, 0)
    assert nat().take_index_error() is True
    assert float(guard[V:].abs().max()) == 0.0
    ok = torch.zeros(V, H, device=DEV)
    keep = (bad >= 0) & (bad < V)
    ok.index_add_(0, bad[keep], d.view(B, S, H)[:, :T][keep].float())
    close(guard[:V], ok, 1e-4, 1e-4, "in-range rows still accumulate")


def test_gather_scatter_colsum_cast():
    B, S, H = 8, 20, 768
    x = rnd(B * S, H); idx = torch.randint(0, S, (B,), device=DEV)
    out = torch.empty(B, H, dtype=torch.bfloat16, device=DEV)
    nat().gather_rows(x, idx, out, B, S, H)
    assert torch.equal(out, x.view(B, S, H)[torch.arange(B), idx])
    dx = torch.zeros_like(x)
    nat().scatter_rows(out, idx, dx, B, S, H)
    ref = torch.zeros_like(x).view(B, S, H); ref[torch.arange(B), idx] = out
    assert torch.equal(dx.view(B, S, H), ref)
    drop = nat().drop_cfg(0.1, 9)
    o2 = torch.empty_like(out); nat().gather_rows(x, idx, o2, B, S, H, drop)
    d2 = torch.zeros_like(x); nat().scatter_rows(out, idx, d2, B, S, H, drop)
    assert torch.equal(o2 == 0, d2.view(B, S, H)[torch.arange(B), idx] == 0)
    # colsum
    for N in (768, 3072, 2304, 100):
        m = rnd(300, N)
        o = torch.empty(N, device=DEV); ws = torch.empty(nat().colsum_ws_floats(N), device=DEV)
        nat().colsum(m, N, 1, 300, 0, N, o, 0.0, ws)
        close(o, m.float().sum(0), 1e-4, 1e-3, "colsum")
    m = rnd(4 * 10, 768)
    o = torch.empty(768, device=DEV); ws = torch.empty(nat().colsum_ws_floats(768), device=DEV)
    nat().colsum(m[3:], 768, 4, 5, 10, 768, o, 0.0, ws)   # rows 3..7 of each group of 10
    close(o, m.view(4, 10, 768)[:, 3:8].float().sum((0, 1)), 1e-4, 1e-3, "colsum groups")
    # casts
    f = rnd(100003, dtype=torch.float32)
    h = torch.empty(100003, dtype=torch.bfloat16, device=DEV)
    nat().cast_f32_to_bf16(f, h)
    assert torch.equal(h, f.to(torch.bfloat16))
    f2 = torch.empty_like(f); nat().cast_bf16_to_f32(h, f2)
    assert torch.equal(f2, h.float())


def test_bce_logits_loss():
    B, N = 32, 3129
    x = rnd(B, N, dtype=torch.float32, scale=3.0).contiguous()
    t = (torch.rand(B, N, device=DEV) > 0.99).float() * 0.6
    loss = torch.empty(1, device=DEV)
    nat().bce_logits_fwd(x, t, loss, B, N)
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(xr, t, reduction="mean") * N
    close(loss[0], ref.detach(), 1e-5, 1e-5, "bce loss")
    ref.backward()
    ldd = 3136
    d = torch.full((B, ldd), 7.0, dtype=torch.bfloat16, device=DEV)
    nat().bce_logits_bwd(x, t, None, d, ldd, B, N)
    close(d[:, :N], xr.grad, 1e-2, 1e-5, "bce grad")
    assert float(d[:, N:].abs().max()) == 0.0


def test_adamw_matches_reference_rules():
    n = 5000
    p0 = rnd(n, dtype=torch.float32); g = rnd(n, dtype=torch.float32)
    seg_end = torch.tensor([2000, 5000], device=DEV); seg_wd = torch.tensor([0.01, 0.0], device=DEV)
    lr, b1, b2, eps = 5e-3, 0.9, 0.999, 1e-8
    # torch.optim.AdamW (mode 1)
    params = [p0[:2000].clone().requires_grad_(True), p0[2000:].clone().requires_grad_(True)]
    opt = torch.optim.AdamW([{"params": [params[0]], "weight_decay": 0.01}, {"params": [params[1]], "weight_decay": 0.0}], lr=lr, betas=(b1, b2), eps=eps)
    p = p0.clone(); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV); p16 = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    for step in (1, 2, 3):
        params[0].grad = g[:2000].clone(); params[1].grad = g[2000:].clone()
        opt.step()
        nat().adamw_step(p, g, m, v, p16, n, seg_end, seg_wd, 2, lr, b1, b2, eps, step, 1, 1, 1.0)
    close(p, torch.cat([params[0].detach(), params[1].detach()]), 1e-5, 1e-6, "adamw torch mode")
    assert torch.equal(p16, p.to(torch.bfloat16))
    # transformers.AdamW rule (mode 0), restated
    p = p0.clone(); m.zero_(); v.zero_()
    pr = p0.clone(); mr = torch.zeros(n, device=DEV); vr = torch.zeros(n, device=DEV)
    wd = torch.cat([torch.full((2000,), 0.01, device=DEV), torch.zeros(3000, device=DEV)])
    for step in (1, 2):
        nat().adamw_step(p, g, m, v, None, n, seg_end, seg_wd, 2, lr, b1, b2, eps, step, 1, 0, 1.0)
        mr = b1 * mr + (1 - b1) * g; vr = b2 * vr + (1 - b2) * g * g
        ss = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        pr = pr - ss * mr / (vr.sqrt() + eps)
        pr = pr - lr * wd * pr
    close(p, pr, 1e-5, 1e-6, "adamw hf mode")


def test_multi_tensor_adamw_optimizer_matches_reference_rules():
    """mmf_amd.modules.optimizers.AdamW (registered "adam_w") vs torch.optim.AdamW (torch_mode) and vs the restated
    transformers.AdamW rule, over 60 tensors of ragged sizes in two weight-decay groups, with gradient clipping."""
    import mmf_amd
    from mmf_amd.common.registry import registry
    cls = registry.get_optimizer_class("adam_w")
    gen = torch.Generator().manual_seed(3)
    shapes = [(int(torch.randint(1, 300, (1,), generator=gen)), int(torch.randint(1, 70, (1,), generator=gen))) for _ in range(59)] + [(30000, 64)]
    base = [torch.randn(s, generator=gen).to(DEV) for s in shapes]
    grads = [torch.randn(s, generator=gen).to(DEV) * 3 for s in shapes]
    def groups(ps):
        return [{"params": ps[0::2], "weight_decay": 0.01}, {"params": ps[1::2], "weight_decay": 0.0}]
    # torch rule
    ours = [b.clone().requires_grad_(True) for b in base]; ref = [b.clone().requires_grad_(True) for b in base]
    o1 = cls(groups(ours), lr=3e-3, eps=1e-8, torch_mode=True); o2 = torch.optim.AdamW(groups(ref), lr=3e-3, eps=1e-8)
    for _ in range(3):
        for p, q, g in zip(ours, ref, grads):
            p.grad = g.clone(); q.grad = g.clone()
        n1 = o1.clip_grad_norm(5.0); n2 = torch.nn.utils.clip_grad_norm_(ref, 5.0)
        close(n1, n2, 1e-5, 1e-5, "grad norm")
        o1.step(); o2.step()
    for p, q in zip(ours, ref):
        close(p.detach(), q.detach(), 2e-5, 2e-6, "adamw multi (torch rule + clip)")
    # transformers rule (restated), no clipping
    ours = [b.clone().requires_grad_(True) for b in base]
    o1 = cls(groups(ours), lr=3e-3, eps=1e-8, correct_bias=True)
    pr = [b.clone() for b in base]; mr = [torch.zeros_like(b) for b in base]; vr = [torch.zeros_like(b) for b in base]
    for step in (1, 2):
        for p, g in zip(ours, grads):
            p.grad = g.clone()
        o1.step()
        for i, g in enumerate(grads):
            wd = 0.01 if i % 2 == 0 else 0.0
            mr[i] = 0.9 * mr[i] + 0.1 * g; vr[i] = 0.999 * vr[i] + 0.001 * g * g
            ss = 3e-3 * math.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
            pr[i] = pr[i] - ss * mr[i] / (vr[i].sqrt() + 1e-8)
            pr[i] = pr[i] - 3e-3 * wd * pr[i]
    for p, q in zip(ours, pr):
        close(p.detach(), q, 2e-5, 2e-6, "adamw multi (transformers rule)")


def test_adamw_reads_bf16_wire_buffers_and_updates_subsets():
    """Round 3: `external_grads` — the data-parallel step hands the optimizer its wire buffers; a bf16 buffer is converted while it
    is loaded (bit-identical to the update from the same values as fp32), ragged tails included — and `step(only=...)` updates a
    subset per call while the step count advances once."""
    from mmf_amd.modules.optimizers import AdamW
    gen = torch.Generator().manual_seed(11)
    shapes = [(40, 33), (257,), (8, 8), (1024, 64), (3,)]
    base = [torch.randn(s, generator=gen).to(DEV) for s in shapes]
    g16 = [(torch.randn(s, generator=gen) * 2).to(DEV).bfloat16() for s in shapes]
    a = [b.clone().requires_grad_(True) for b in base]; b_ = [b.clone().requires_grad_(True) for b in base]
    oa = AdamW([{"params": a, "weight_decay": 0.01}], lr=2e-3, eps=1e-8); ob = AdamW([{"params": b_, "weight_decay": 0.01}], lr=2e-3, eps=1e-8)
    oa.external_grads = {id(p): g for p, g in zip(a, g16)}                       # bf16 buffers, read in place
    for step in range(2):
        for p, g in zip(b_, g16):
            p.grad = g.float()
        ob.step()
        oa.step(only={id(p) for p in a[:2]}, advance=True)                      # two calls = one optimizer step
        oa.step(only={id(p) for p in a[2:]}, advance=False)
    for p, q in zip(a, b_):
        assert p.grad is None
        assert torch.equal(p.detach(), q.detach())
    assert all(oa.state[p]["step"] == 2 for p in a)
    with pytest.raises((nat().NativeLibraryError, RuntimeError)):      # (the C ABI's error through ctypes, or the operator library's own check: torch_ops.cpp _adamw_step)
        oa.external_grads = {id(a[0]): g16[0].half()}
        oa.step()


@pytest.mark.parametrize("capturable", [False, True])
def test_optimizer_checkpoint_resumes_step_count_and_schedule(capturable):
    """state_dict() carries a per-parameter `step` (the transformers.AdamW format) and, for the capturable form, the device
    words that hold the live step count / warm-up factor: 2 updates + checkpoint + 2 updates in a NEW optimizer equal 4
    uninterrupted updates; a reference-format checkpoint (no device words) resumes at its step count; a parameter that
    gets its first gradient late starts its own bias correction at step 1."""
    from mmf_amd.modules.optimizers import AdamW
    gen = torch.Generator().manual_seed(9)
    base = [torch.randn(40, 33, generator=gen).to(DEV), torch.randn(257, generator=gen).to(DEV), torch.randn(8, 8, generator=gen).to(DEV)]
    grads = [[torch.randn(b.shape, generator=gen).to(DEV) for b in base] for _ in range(4)]
    kw = dict(lr=2e-3, weight_decay=0.01, capturable=capturable)
    if capturable:
        kw["schedule"] = ("warmup_linear", 3.0, 10.0)

    def run(opt, ps, steps):
        for gs in steps:
            for p, g in zip(ps, gs):
                p.grad = g.clone()
            opt.step()

    a = [b.clone().requires_grad_(True) for b in base]
    oa = AdamW(a, **kw)
    run(oa, a, grads)
    b = [x.clone().requires_grad_(True) for x in base]
    ob = AdamW(b, **kw)
    run(ob, b, grads[:2])
    ck = ob.state_dict()
    assert all(int(st["step"]) == 2 for st in ck["state"].values())
    assert ("mmf_amd_dev_state" in ck) == capturable
    # a checkpoint in the reference optimizer's format (per-parameter step, no device words); cloned now: torch's
    # load_state_dict keeps tensors that already have the right dtype / device, so `oc` below shares `ck`'s moments
    ref_ck = {"state": {i: {"step": 2, "exp_avg": st["exp_avg"].clone(), "exp_avg_sq": st["exp_avg_sq"].clone()} for i, st in ck["state"].items()},
              "param_groups": ck["param_groups"]}
    import copy
    c = [x.detach().clone().requires_grad_(True) for x in b]
    oc = AdamW(c, **kw)
    oc.load_state_dict(copy.deepcopy(ck))
    run(oc, c, grads[2:])
    for p, q in zip(a, c):
        close(q.detach(), p.detach(), 1e-6, 1e-7, "resumed == uninterrupted")
    d = [x.detach().clone().requires_grad_(True) for x in b]
    od = AdamW(d, **kw)
    od.load_state_dict(ref_ck)
    run(od, d, grads[2:])
    for p, q in zip(a, d):
        close(q.detach(), p.detach(), 1e-6, 1e-7, "resumed from a reference-format checkpoint")
    if not capturable:
        # late first gradient: its own step count starts at 1 (bias correction of a fresh Adam state)
        e = [x.clone().requires_grad_(True) for x in base]
        oe = AdamW(e, lr=2e-3, weight_decay=0.0)
        for k in range(2):
            e[0].grad = grads[k][0].clone(); e[1].grad = grads[k][1].clone(); e[2].grad = None
            oe.step()
        e[2].grad = grads[2][2].clone(); e[0].grad = None; e[1].grad = None
        oe.step()
        f = [base[2].clone().requires_grad_(True)]
        of = AdamW(f, lr=2e-3, weight_decay=0.0)
        f[0].grad = grads[2][2].clone()
        of.step()
        close(e[2].detach(), f[0].detach(), 1e-6, 1e-7, "late parameter")
        assert oe.state[e[2]]["step"] == 1 and oe.state[e[0]]["step"] == 2


def test_optimizer_refreshes_bf16_shadows_in_place():
    """After a fused step the cached bf16 weight shadow equals bf16(new fp32 master) without a re-cast."""
    from mmf_amd import functional as Fn
    from mmf_amd.modules.hf_layers import Linear
    from mmf_amd.modules.optimizers import AdamW
    lin = Linear(64, 128).to(DEV)
    torch.nn.init.normal_(lin.weight); torch.nn.init.normal_(lin.bias)
    x = rnd(8, 64)
    y = lin(x); y.float().sum().backward()
    sh = Fn.shadows.get(lin.weight)
    opt = AdamW(lin.parameters(), lr=1e-2)
    opt.step()
    assert Fn.shadows.get(lin.weight) is sh              # still considered fresh (no new cast)
    assert torch.equal(sh, lin.weight.detach().to(torch.bfloat16))


def test_optimizer_keeps_packed_qkv_biases_current():
    """Q|K|V biases are consumed from one packed fp32 copy; the fused step must write it too, or the forward would keep
    using the biases of step 0."""
    from mmf_amd import functional as Fn
    from mmf_amd.modules.hf_layers import BertConfig, BertSelfAttentionJit
    from mmf_amd.modules.optimizers import AdamW
    att = BertSelfAttentionJit(BertConfig(hidden_size=128, num_attention_heads=2)).to(DEV)
    for p in att.parameters():
        torch.nn.init.normal_(p, std=0.05)
    x = rnd(2, 16, 128)
    ctx, _ = att(x)
    ctx.float().sum().backward()
    w16, b32 = att.packed_qkv()
    before = b32.clone()
    AdamW(att.parameters(), lr=1e-2).step()
    w16b, b32b = att.packed_qkv()
    assert b32b is b32 and w16b is w16                    # no re-pack
    expect = torch.cat([att.query.bias, att.key.bias, att.value.bias]).detach()
    assert torch.equal(b32, expect) and not torch.equal(b32[256:], before[256:])
    assert torch.equal(w16, torch.cat([att.query.weight, att.key.weight, att.value.weight]).detach().bfloat16())


def test_capturable_adamw_matches_host_counters_and_warmup_linear():
    from mmf_amd.modules.optimizers import AdamW
    torch.manual_seed(0)
    ps = [torch.randn(300, 7, device=DEV), torch.randn(64, device=DEV)]
    pa = [torch.nn.Parameter(p.clone()) for p in ps]; pb = [torch.nn.Parameter(p.clone()) for p in ps]
    oa = AdamW(pa, lr=1e-2, weight_decay=0.01)
    ob = AdamW(pb, lr=1e-2, weight_decay=0.01, capturable=True)
    for it in range(4):
        gs = [torch.randn_like(p) for p in ps]
        for q, r, g in zip(pa, pb, gs):
            q.grad = g.clone(); r.grad = g.clone()
        oa.step(); ob.step()
    for q, r in zip(pa, pb):
        assert torch.allclose(q, r, rtol=1e-6, atol=1e-7)
    assert float(ob._dev_state[0]) == 4.0
    # device-side warmup_linear == lr * lambda(t - 1) applied by hand
    pc = [torch.nn.Parameter(p.clone()) for p in ps]; pd = [torch.nn.Parameter(p.clone()) for p in ps]
    oc = AdamW(pc, lr=1e-2, capturable=True, schedule=("warmup_linear", 2, 6))
    od = AdamW(pd, lr=1e-2)
    lam = lambda s: s / 2 if s < 2 else max(0.0, (6 - s) / 4)
    for it in range(5):
        gs = [torch.randn_like(p) for p in ps]
        for q, r, g in zip(pc, pd, gs):
            q.grad = g.clone(); r.grad = g.clone()
        for grp in od.param_groups:
            grp["lr"] = 1e-2 * lam(it)
        oc.step(); od.step()
    for q, r in zip(pc, pd):
        assert torch.allclose(q, r, rtol=1e-5, atol=1e-7)


# ---- skinny problems: K-slices over the chip, the fused epilogue runs on the slab sums (mmf_gemm_skinny_splits) -------------------------
NO_SKINNY = 1 << 18


@pytest.mark.parametrize("M", [32, 8, 64])
def test_gemm_skinny_head_shapes_match_torch_and_the_unsplit_launch(M):
    """The classification head's GEMMs (visual_bert.py:349-404: dense + GELU, classifier 768 -> 3129) and their input gradients at
    batch-sized M: split over ~128 workgroups, epilogue applied after the slab sum; against fp32 torch and against the same call
    without the workspace (one workgroup per tile walking the whole reduction)."""
    H, L = 768, 3129
    assert nat().lib().mmf_gemm_skinny_splits(M, H, L, 0) > 1 and nat().lib().mmf_gemm_skinny_splits(M, H, 2048, 0) > 1
    assert nat().lib().mmf_gemm_skinny_splits(M, H, H, 0) == 1          # (short reductions stay one launch)
    assert nat().lib().mmf_gemm_skinny_splits(7296, H, L, 0) == 1 and nat().lib().mmf_gemm_skinny_splits(L, H, 32, 1) == 1
    KD = 2048
    x = rnd(M, KD, seed=1); Wd = rnd(H, KD, seed=2, scale=0.03); bd = rnd(H, dtype=torch.float32, seed=3)
    # dense + exact-erf GELU with the saved derivative over a long reduction
    h = torch.empty(M, H, dtype=torch.bfloat16, device=DEV); u = torch.empty_like(h)
    nat().gemm(x, Wd, h, M, H, KD, KD, KD, H, bias=bd, act=1, U=u)
    assert "skinny" in nat().gemm_last_kernel()
    pre = x.float() @ Wd.float().t() + bd
    close(h, torch.nn.functional.gelu(pre), 1e-2, 2e-2, "skinny dense + gelu")
    h2 = torch.empty_like(h); u2 = torch.empty_like(u)
    nat().gemm(x, Wd, h2, M, H, KD, KD, KD, H, bias=bd, act=1, U=u2, debug_flags=NO_SKINNY)
    assert "skinny" not in nat().gemm_last_kernel()
    close(h, h2, 1e-2, 1e-2, "skinny vs unsplit"); close(u, u2, 1e-2, 1e-2, "saved gelu'")
    # 3129 outputs (not a multiple of 8), fp32 scores
    Wc = rnd(L, H, seed=4, scale=0.05); bc = rnd(L, dtype=torch.float32, seed=5)
    Wl = rnd(L, KD, seed=8, scale=0.03)
    sc = torch.full((M, L), float("nan"), device=DEV)
    nat().gemm(x, Wl, sc, M, L, KD, KD, KD, L, bias=bc)
    assert "skinny" in nat().gemm_last_kernel()
    close(sc, x.float() @ Wl.float().t() + bc, 1e-3, 3e-3, "skinny wide classifier")
    # input gradient of the classifier: reduction over the 3129 labels (49 K-steps), k-major weight, bf16 output
    LP = 3136                                   # the row operand's leading dimension covers round_up(K, 8), zero padding
    dsc_p = torch.zeros(M, LP, dtype=torch.bfloat16, device=DEV); dsc_p[:, :L] = rnd(M, L, seed=6)
    dsc = dsc_p[:, :L]
    dh = torch.empty(M, H, dtype=torch.bfloat16, device=DEV)
    nat().gemm(dsc_p, Wc, dh, M, H, L, LP, H, H, b_kmajor=True)
    assert "skinny" in nat().gemm_last_kernel()
    close(dh, dsc.float() @ Wc.float(), 1e-2, 2e-2, "skinny classifier dgrad")
    # ... times the saved derivative, plus a residual gradient, with dropout (mask = hash of the element index: same with and without)
    res = rnd(M, H, seed=7)
    d1 = torch.empty_like(dh); d2 = torch.empty_like(dh)
    drop = nat().drop_cfg(0.1, 4321, None)
    nat().gemm(dsc_p, Wc, d1, M, H, L, LP, H, H, b_kmajor=True, act=2, aux=u, drop=drop)
    nat().gemm(dsc_p, Wc, d2, M, H, L, LP, H, H, b_kmajor=True, act=2, aux=u, drop=drop, debug_flags=NO_SKINNY)
    close(d1, d2, 1e-2, 1e-2, "skinny act-2 + dropout vs unsplit")
    nat().gemm(dsc_p, Wc, d1, M, H, L, LP, H, H, b_kmajor=True, resid=res, ldr=H)
    close(d1, dsc.float() @ Wc.float() + res.float(), 1e-2, 2e-2, "skinny dgrad + residual")


def test_gemm_skinny_row_remap_and_decode_opt_out():
    """The remaining epilogue of `splitk_epilogue_kernel`: the row remap `grp` (output row = m + (m / R) * skip + row0, what the
    decoding cache writes use) through the skinny path against the one-kernel launch, bit for bit on the rows written and nothing else touched;
    and the incremental M4C decoder's K = 3072 GEMM opts out of the path (mmf_amd/modules/infer.py: one launch per call in a host-bound loop)."""
    M, N, K, R, skip, row0 = 16, 768, 3072, 2, 5, 3
    assert nat().lib().mmf_gemm_skinny_splits(M, N, K, 0) > 1
    x = rnd(M, K, seed=11); W = rnd(N, K, seed=12, scale=0.03); bias = rnd(N, dtype=torch.float32, seed=13)
    rows = (M // R) * (R + skip) + row0
    outs = []
    for flags in (0, NO_SKINNY):
        out = torch.full((rows, N), 7.0, dtype=torch.bfloat16, device=DEV)
        nat().gemm(x, W, out, M, N, K, K, K, N, bias=bias, grp=(R, skip, row0), debug_flags=flags)
        assert ("skinny" in nat().gemm_last_kernel()) == (flags == 0)
        outs.append(out)
    ref = x.float() @ W.float().t() + bias
    dst = torch.tensor([m + (m // R) * skip + row0 for m in range(M)], device=DEV)
    close(outs[0][dst], ref, 1e-2, 2e-2, "skinny + row remap")
    close(outs[0][dst], outs[1][dst], 1e-2, 1e-2, "skinny vs unsplit, remapped rows")
    untouched = torch.ones(rows, dtype=torch.bool, device=DEV); untouched[dst] = False
    assert bool((outs[0][untouched] == 7.0).all()) and bool((outs[1][untouched] == 7.0).all())
    import inspect
    from mmf_amd.modules import infer
    assert "GEMM_NO_SKINNY" in inspect.getsource(infer.layer_rows)


def test_pack_f32_multi_packs_scales_and_converts_in_one_launch():
    """The data-parallel reducer's bucket packing (mmf_amd/trainers/core/device.py): many fp32 gradients -> one flat fp32 / bf16 buffer at given
    offsets, scaled by 1 / world before the rounding; ragged sizes, more tensors than one launch holds, untouched gaps."""
    g = torch.Generator().manual_seed(5)
    sizes = [768, 768 * 768, 5, 3129, 64, 1, 30522 * 4 + 3] + [96] * 60
    ts = [torch.randn(n, generator=g).to(DEV) for n in sizes]
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 63) // 64 * 64
    for dtype in (torch.float32, torch.bfloat16):
        flat = torch.full((total,), 7.0, dtype=dtype, device=DEV)
        nat().pack_f32_multi(ts, offs, flat, 0.5)
        seen = torch.zeros(total, dtype=torch.bool, device=DEV)
        for t, o in zip(ts, offs):
            assert torch.equal(flat[o:o + t.numel()], (t * 0.5).to(dtype))
            seen[o:o + t.numel()] = True
        assert bool((flat[~seen] == 7.0).all())


def test_gemm_skinny_repeated_launches_are_stable():
    M, H, L = 32, 768, 3129
    dsc = torch.zeros(M, 3136, dtype=torch.bfloat16, device=DEV); dsc[:, :L] = rnd(M, L, seed=6)
    Wc = rnd(L, H, seed=4, scale=0.05)
    dh = torch.empty(M, H, dtype=torch.bfloat16, device=DEV)
    nat().gemm(dsc, Wc, dh, M, H, L, 3136, H, H, b_kmajor=True)
    first = dh.clone()
    for _ in range(20):
        nat().gemm(dsc, Wc, dh, M, H, L, 3136, H, H, b_kmajor=True)
    assert torch.equal(dh, first)
