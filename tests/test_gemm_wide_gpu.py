"""The wide-tile forward GEMM (mmf_amd/csrc/gemm_wide.h: one 256x96 / 192x192 / 256x128 tile per CU, ping-pong wave groups over an
LDS-DMA ring) against fp32 torch and against the 128-row kernel on the same inputs — bit-identical, both add the K-steps in
the same order — for every epilogue the encoder uses, ragged M, short and long K, plus a repeated-launch race screen."""
import pytest
import torch

from tests.test_kernels_gpu import DEV, close, nat, rnd

pytestmark = pytest.mark.gpu
NO_WIDE = 1 << 17          # debug_flags: never a wide tile
NO_KSPLIT = 1 << 13        # keep the 128-row kernel on the plain K order
CONFIGS = {1: (256, 96), 2: (192, 192), 3: (256, 128), 4: (128, 96), 5: (128, 128), 6: (192, 96)}


@pytest.fixture
def force_wide():
    def set_(cfg):
        nat().set_tunable(nat().TUN_GEMM_WIDE, cfg)
    yield set_
    nat().set_tunable(nat().TUN_GEMM_WIDE, 0)


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("M,K", [(7296, 768), (1500, 768), (512, 128), (3200, 3072), (2000, 192), (3968, 2304)])
def test_wide_forward_bias_matches_torch_and_the_128_row_kernel(cfg, M, K, force_wide):
    N = {1: 768, 2: 2304, 3: 3072, 4: 768, 5: 1024, 6: 768}[cfg] if M > 2000 else {1: 192, 2: 384, 3: 512, 4: 192, 5: 256, 6: 192}[cfg]
    A = rnd(M, K, seed=1); B = rnd(N, K, seed=2, scale=0.05); bias = torch.randn(N, device=DEV)
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV); C0 = torch.empty_like(C)
    force_wide(cfg)
    nat().gemm(A, B, C, M, N, K, K, K, N, bias=bias)
    nat().gemm(A, B, C0, M, N, K, K, K, N, bias=bias, debug_flags=NO_WIDE | NO_KSPLIT)
    close(C, A.float() @ B.float().t() + bias, 1e-2, 2e-2, "wide tile %s forward" % (CONFIGS[cfg],))
    assert torch.equal(C, C0)


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6])
def test_wide_epilogues(cfg, force_wide):
    M, K = 1500, 768
    N = {1: 768, 2: 768, 3: 1024, 4: 768, 5: 1024, 6: 768}[cfg]
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


def test_few_token_rows_take_the_128_row_wide_tile(force_wide):
    """A trimmed batch (32 x 124 = 3968 token rows): the N = 768 sites run on the 128 x 96 ping-pong tile (248 workgroups instead of 128), the wide
    outputs and the full-length batch keep their tiles."""
    for M, N, K, kernel in ((3968, 768, 3072, "gemm_wide_kernel 128x96"), (3968, 768, 768, "gemm_wide_kernel 128x96"),
                            (7296, 768, 3072, "gemm_wide_kernel 256x96"), (5248, 768, 768, "gemm_wide_kernel 192x96"), (6144, 768, 3072, "gemm_wide_kernel 192x96"),
                            (3200, 1024, 1024, "gemm_wide_kernel 128x128"), (4096, 1024, 4096, "gemm_wide_kernel 128x128"),
                            (14592, 1024, 1024, "gemm_persist_kernel 256x128")):
        A = rnd(M, K, seed=8); B = rnd(N, K, seed=9, scale=0.05)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); C0 = torch.empty_like(C)
        force_wide(0)
        nat().gemm(A, B, C, M, N, K, K, K, N)
        assert nat().gemm_last_kernel() == kernel, (M, N, K, nat().gemm_last_kernel())
        nat().gemm(A, B, C0, M, N, K, K, K, N, debug_flags=NO_WIDE | NO_KSPLIT)
        assert torch.equal(C, C0)


@pytest.mark.parametrize("cfg,N", [(1, 768), (2, 2304), (3, 3072), (4, 768), (5, 1024), (6, 768)])
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


# ---- weight-gradient form (both operands k-major) on the wide tile: the grouped launch of a layer's four weight gradients ----------
@pytest.fixture
def wgrad_tun():
    def set_(v):
        nat().set_tunable(nat().TUN_WGRAD_WIDE, v)
    yield set_
    nat().set_tunable(nat().TUN_WGRAD_WIDE, 0)


def _layer_wgrad_problems(T, H, I, with_db, fill=float("nan")):
    du = rnd(T, I, seed=1); a_out = rnd(T, H, seed=2); dlin2 = rnd(T, H, seed=3); hh = rnd(T, I, seed=4)
    dqkv = rnd(T, 3 * H, seed=5); x = rnd(T, H, seed=6); dlin1 = rnd(T, H, seed=7); ctxt = rnd(T, H, seed=8)
    specs = [(du, a_out, I, H), (dlin2, hh, H, I), (dqkv, x, 3 * H, H), (dlin1, ctxt, H, H)]
    probs, outs = [], []
    for dy, xx, N, K in specs:
        dw = torch.full((N, K), fill, device=DEV)
        db = torch.full((N,), fill, device=DEV) if with_db else None
        probs.append(dict(A=dy, B=xx, C_out=dw, M=N, N=K, K=T, lda=N, ldb=K, ldc=K, a_kmajor=True, b_kmajor=True, rowsum_out=db))
        outs.append((dw, db))
    return specs, probs, outs


@pytest.mark.parametrize("T", [7296, 640, 192])
@pytest.mark.parametrize("with_db", [True, False])
def test_wide_grouped_weight_gradients_match_torch_and_the_128_row_tiles(T, with_db, wgrad_tun):
    """dW = dY^T X (+ db = column sums of dY) for the four GEMMs of a layer on 256 x 128 tiles (k-major LDS images read with the
    transpose read, three-stage ring) against fp32 torch and against the 128 x 128 grouped launch: with the bias gradient riding along
    both add the 32-deep K sub-steps in the same order into fp32 accumulators, so the results are bit-identical."""
    import math
    H, I = 768, 3072
    specs, probs, outs = _layer_wgrad_problems(T, H, I, with_db)
    wgrad_tun(2)
    nat().gemm_grouped(probs)
    assert "wide_grouped" in nat().gemm_last_kernel()
    specs2, probs2, outs2 = _layer_wgrad_problems(T, H, I, with_db)
    wgrad_tun(1)
    nat().gemm_grouped(probs2)
    assert "wide" not in nat().gemm_last_kernel()
    for (dy, xx, N, K), (dw, db), (dw2, db2) in zip(specs, outs, outs2):
        close(dw, dy.float().t() @ xx.float(), 1e-4, 2e-3 * math.sqrt(T) / 8, "wide grouped dW %dx%d" % (N, K))
        if with_db:      # (without the bias gradient the 128-row launch splits every K-step between two wave groups: another fp32 summation order)
            assert torch.equal(dw, dw2), "wide vs 128-row tiles, dW %dx%d: %g" % (N, K, float((dw - dw2).abs().max()))
        else:
            close(dw, dw2, 1e-5, 2e-4 * math.sqrt(T) / 8, "wide vs 128-row tiles")
        if with_db:
            close(db, dy.float().sum(0), 1e-5, 1e-3 * math.sqrt(T) / 8, "wide grouped bias gradient")
            assert torch.equal(db, db2)


def test_wide_grouped_weight_gradients_accumulate_with_beta(wgrad_tun):
    """beta = 1 (gradient accumulation): the fp32 output is read back and added in the epilogue."""
    T, H = 1280, 768
    dy = rnd(T, 2 * H, seed=11); x = rnd(T, H, seed=12)
    base = torch.randn(2 * H, H, device=DEV)
    dw = base.clone()
    wgrad_tun(2)
    nat().gemm_grouped([dict(A=dy, B=x, C_out=dw, M=2 * H, N=H, K=T, lda=2 * H, ldb=H, ldc=H, a_kmajor=True, b_kmajor=True, beta=1.0)])
    assert "wide_grouped" in nat().gemm_last_kernel()
    close(dw, base + dy.float().t() @ x.float(), 1e-4, 2e-3, "accumulated dW")


def test_wide_grouped_falls_back_when_a_problem_is_not_whole_tiles(wgrad_tun):
    T, H = 640, 768
    dy = rnd(T, 384, seed=21); x = rnd(T, H, seed=22)         # 384 output rows: not a multiple of 256
    dw = torch.empty(384, H, device=DEV)
    wgrad_tun(2)
    nat().gemm_grouped([dict(A=dy, B=x, C_out=dw, M=384, N=H, K=T, lda=384, ldb=H, ldc=H, a_kmajor=True, b_kmajor=True)])
    assert "wide" not in nat().gemm_last_kernel()
    close(dw, dy.float().t() @ x.float(), 1e-4, 2e-3, "fallback dW")


def test_wide_grouped_repeated_launches_are_stable(wgrad_tun):
    """Race screen of the k-major ring (LDS-DMA pieces against transpose reads): 20 launches, identical bits."""
    specs, probs, outs = _layer_wgrad_problems(7296, 768, 3072, True)
    wgrad_tun(0)
    nat().gemm_grouped(probs)
    assert "wide_grouped" in nat().gemm_last_kernel()      # the default dispatch takes the wide tile for the encoder layer
    first = [(dw.clone(), db.clone()) for dw, db in outs]
    for _ in range(20):
        nat().gemm_grouped(probs)
    for (dw, db), (f_dw, f_db) in zip(outs, first):
        assert torch.equal(dw, f_dw) and torch.equal(db, f_db)


def test_deferred_weight_gradients_join_grouped_launches():
    """functional.wgrad_defer (round 4; active inside the graphed steps): the weight gradients of the autograd nodes that are not the fused encoder
    layer are queued and launched eight at a time as ONE grouped GEMM — per HIP stream — instead of one split-K GEMM + slab reduction each.
    Nothing is written before the flush; afterwards dW / db equal the immediate launches up to fp32 summation order."""
    from mmf_amd import functional as Fn
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(3)
    shapes = [(4096, 1024, 1024, True), (3232, 1024, 1024, False), (4096, 768, 3072, True), (3232, 1024, 768, True), (4096, 3072, 768, False),
              (4096, 1024, 1024, True), (4096, 256, 128, True), (4096, 1024, 1024, False), (4096, 768, 768, True), (4096, 1024, 1024, True)]      # (tokens, out, in, bias)
    ops = []
    for M, N, K, want_db in shapes:
        dy = (torch.randn(M, N, generator=g) * 0.1).bfloat16().to(dev)
        x = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(dev)
        w16 = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(dev)
        ops.append((dy, x, w16, M, N, K, want_db))
    ref = [Fn._linear_bwd(dy, N, x, w16, M, N, K, need_dx=False, want_db=want_db) for dy, x, w16, M, N, K, want_db in ops]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    got = []
    with Fn.wgrad_defer():
        assert Fn.wgrad_defer.active
        for i, (dy, x, w16, M, N, K, want_db) in enumerate(ops):
            if i == 3:      # one problem from another stream: its own queue
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    got.append(Fn._linear_bwd(dy, N, x, w16, M, N, K, need_dx=False, want_db=want_db, defer=True))
                torch.cuda.current_stream().wait_stream(side)
            else:
                got.append(Fn._linear_bwd(dy, N, x, w16, M, N, K, need_dx=False, want_db=want_db, defer=True))
        queued = sum(len(q) for q in Fn.wgrad_defer.queues.values())
        assert 0 < queued < len(ops) and len(Fn.wgrad_defer.queues) >= 3      # wide-tile queue (launched once at eight), 128-row queue, the side stream's
    assert not Fn.wgrad_defer.active and not Fn.wgrad_defer.queues              # leaving the block launched what was left
    torch.cuda.synchronize()
    for r, o, (dy, x, w16, M, N, K, want_db) in zip(ref, got, ops):
        assert (o[0] is None) and (r[0] is None)
        scale = float(r[1].abs().max())
        assert float((o[1] - r[1]).abs().max()) <= 2e-5 * scale + 1e-6, (M, N, K)
        if want_db:
            assert float((o[2] - r[2]).abs().max()) <= 2e-5 * float(r[2].abs().max()) + 1e-5, (M, N, K)
    # round 5 (ADVICE, high): deferral is opt-in per call (`defer=True`: encoder-internal matrices with ONE gradient contribution per backward);
    # a node that does not opt in — LinearFn, the vocabulary heads, M4C's scores — launches immediately even inside the block, and the SAME
    # weight a second time is never queued twice (autograd would sum an unfilled buffer)
    dy, x, w16, M, N, K, want_db = ops[0]
    with Fn.wgrad_defer():
        Fn._linear_bwd(dy, N, x, w16, M, N, K, need_dx=False, want_db=want_db)
        assert not any(Fn.wgrad_defer.queues.values())
        first = Fn._linear_bwd(dy, N, x, w16, M, N, K, need_dx=False, want_db=want_db, defer=True)
        assert sum(len(q) for q in Fn.wgrad_defer.queues.values()) == 1
        second = Fn._linear_bwd(dy, N, x, w16, M, N, K, need_dx=False, want_db=want_db, defer=True)       # same weight again: flush, then immediate
        assert not any(Fn.wgrad_defer.queues.values())
        torch.cuda.synchronize()
        assert torch.equal(first[1], ref[0][1]) or float((first[1] - ref[0][1]).abs().max()) <= 2e-5 * float(ref[0][1].abs().max()) + 1e-6
        assert torch.equal(second[1], ref[0][1])
