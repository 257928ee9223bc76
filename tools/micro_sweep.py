"""On-hardware sweeps of the tuning knobs (include/mmf_amd.h MMF_TUN_*): LayerNorm-backward grid, split-K count of the
weight-gradient GEMMs (both GEMM forms).  python tools/micro_sweep.py [ln] [wgrad] [attn]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmf_amd import _native as nat


def timeit(f, iters=30):
    """us per call of `f`, measured on a hipGraph of `iters` back-to-back calls (no host launch floor)."""
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        f()
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                f()
    torch.cuda.synchronize()
    g.replay()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e3


def ln_sweep():
    rows, H = int(os.environ.get("LN_ROWS", "7296")), int(os.environ.get("LN_H", "768"))      # (ViLBERT's visual stream: LN_ROWS=3232 LN_H=1024)
    dev = "cuda"
    dy = torch.randn(rows, H, device=dev).bfloat16(); x = torch.randn(rows, H, device=dev).bfloat16()
    mean = torch.randn(rows, device=dev); rstd = torch.rand(rows, device=dev) + 0.5; gamma = torch.randn(H, device=dev)
    dx = torch.empty_like(dy); dlin = torch.empty_like(dy)
    dg = torch.empty(H, device=dev); db = torch.empty(H, device=dev); dbias = torch.empty(H, device=dev)
    ws = torch.empty(nat.layernorm_bwd_ws_floats(H), device=dev)
    y = torch.empty_like(x)
    for old in (0, 2, 1):      # 0: half-wave kernels, two rows in flight; 2: the same, one row in flight; 1: the one-wave-per-row kernels
        nat.set_tunable(nat.TUN_ALT_FORMS, {0: 0, 2: 8, 1: 1}[old])
        tf = timeit(lambda: nat.layernorm_fwd(x, gamma, gamma, y, mean, rstd, rows, H, 1e-12))
        t0 = timeit(lambda: nat.layernorm_bwd(dy, x, mean, rstd, gamma, dx, None, nat.NO_DROP, dg, db, None, False, ws, rows, H))
        t1 = timeit(lambda: nat.layernorm_bwd(dy, x, mean, rstd, gamma, dx, dlin, nat.drop_cfg(0.1, 5), dg, db, None, False, ws, rows, H))
        t2 = timeit(lambda: nat.layernorm_bwd(dy, x, mean, rstd, gamma, dx, dlin, nat.drop_cfg(0.1, 5), dg, db, dbias, False, ws, rows, H))
        print("%s kernels: fwd %5.1f us (%.2f TB/s)  bwd plain %5.1f  bwd+dropout %5.1f (%.2f TB/s incl. reduce)  +dbias %5.1f" % (
            {0: "half-wave-per-row", 2: "half-wave, 1 row  ", 1: "one-wave-per-row "}[old], tf, 4.0 * rows * H / tf / 1e6, t0, t1, 8.0 * rows * H / t1 / 1e6, t2), flush=True)
    nat.set_tunable(nat.TUN_ALT_FORMS, 0)


def attn_sweep():
    """Attention kernels at the VisualBERT VQA2 shape, steady state (hipGraph replays): forward, one-pass backward, two-kernel backward."""
    B, A, S, H = 32, 12, 228, 768
    qkv = (torch.randn(B * S, 3 * H, device="cuda") * 0.5).bfloat16()
    mask = torch.zeros(B, S, device="cuda")
    ctx = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16); ctx32 = torch.empty(B * S, H, device="cuda")
    lse = torch.empty(B * A * S, device="cuda"); delta = torch.empty(B * A * S, device="cuda")
    dctx = (torch.randn(B * S, H, device="cuda") * 0.1).bfloat16(); dqkv = torch.empty_like(qkv)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    for p in (0.1, 0.0):
        drop = nat.drop_cfg(p, 99, None) if p else nat.NO_DROP
        for o32 in (ctx32, None):
            fwd = lambda: nat.attention_fwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, A, S, S, 0.125, drop=drop, ctx_f32=o32)
            bwd = lambda: nat.attention_bwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, A, S, S, 0.125, dctx, dqkv[:, :H],
                                            dqkv[:, H:2 * H], dqkv[:, 2 * H:], delta, drop=drop, ctx_f32=o32)
            t_f = timeit(fwd)
            nat.set_tunable(nat.TUN_ALT_FORMS, 0); t_1 = timeit(bwd)
            nat.set_tunable(nat.TUN_ALT_FORMS, 4); t_2 = timeit(bwd)
            nat.set_tunable(nat.TUN_ALT_FORMS, 0)
            print("attention B=32 S=228 dropout %.1f fp32-ctx %-5s: fwd %6.1f us   bwd one-pass %6.1f us   bwd two-kernel %6.1f us"
                  % (p, o32 is not None, t_f, t_1, t_2), flush=True)
            if p:      # the forward hands its dropout decisions to the one-pass backward as a bit table (mmf_attn_desc.keep_bits)
                kb = torch.empty(nat.attention_keep_bits_words(B, A, S, S, 64), dtype=torch.int32, device="cuda")
                t_fk = timeit(lambda: nat.attention_fwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, A, S, S, 0.125, drop=drop, ctx_f32=o32, keep_bits=kb))
                t_bk = timeit(lambda: nat.attention_bwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, A, S, S, 0.125, dctx, dqkv[:, :H],
                                                        dqkv[:, H:2 * H], dqkv[:, 2 * H:], delta, drop=drop, ctx_f32=o32, keep_bits=kb))
                print("          ... with the keep-bit table      : fwd %6.1f us   bwd one-pass %6.1f us" % (t_fk, t_bk), flush=True)


ABLATIONS = (("default", 0), ("no-dma", 1 << 4), ("no-mfma", 2 << 4), ("no-epilogue", 8 << 4), ("dma-only", (2 | 8) << 4),
             ("mfma+lds-only", (1 | 8) << 4), ("epilogue-only", (1 | 2) << 4), ("skeleton", (1 | 2 | 8) << 4))


def gemm_variants(variants=(("default", 0), ("ksplit", 4096), ("no-ksplit", 8192))):
    """All twelve per-layer GEMMs: default dispatch, K-split wave layout forced (bit 12) and forbidden (bit 13); or, with
    ABLATIONS, the kernel with parts switched off (debug_flags bits 4-7: bit 4 no operand prefetch after the first stage,
    bit 5 no LDS reads / MFMAs, bit 7 no epilogue) - results are garbage, only the timings mean something."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from gemm_bench import SHAPES
    dev = "cuda"
    tot = {n: 0.0 for n, _ in variants}
    for name, kind, m, n, k in SHAPES:
        if kind == "NT":
            A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16()
            C = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            f = lambda fl: nat.gemm(A, B, C, m, n, k, k, k, n, debug_flags=fl)
        elif kind == "NN":
            A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16()
            C = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            f = lambda fl: nat.gemm(A, B, C, m, n, k, k, n, n, b_kmajor=True, debug_flags=fl)
        else:
            A = torch.randn(k, m, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16()
            C = torch.empty(m, n, device=dev, dtype=torch.float32)
            f = lambda fl: nat.gemm(A, B, C, m, n, k, m, n, n, a_kmajor=True, b_kmajor=True, debug_flags=fl)
        res = {}
        for rep in range(2):
            for vn, fl in variants:
                res.setdefault(vn, []).append(timeit(lambda: f(fl)))
        line = "%-11s %s N=%5d K=%5d " % (name, kind, n, k)
        for vn, _ in variants:
            t = min(res[vn]); tot[vn] += t
            line += " %s %5.1f us (%4.0f TF)" % (vn, t, 2.0 * m * n * k / t / 1e6)
        print(line, flush=True)
    print("layer totals: " + "  ".join("%s %.1f us" % (vn, tot[vn]) for vn, _ in variants))


if __name__ == "__main__":
    which = sys.argv[1:] or ["ln"]
    if "ln" in which:
        ln_sweep()
    if "gemm" in which:
        gemm_variants()
    if "ablate" in which:
        gemm_variants(ABLATIONS)
    if "wide" in which:       # forward shapes: dispatcher's choice (wide tiles where the model says so) against the 128-row kernel
        gemm_variants((("auto", 0), ("128-row", 1 << 17)))
    if "attn" in which:
        attn_sweep()
