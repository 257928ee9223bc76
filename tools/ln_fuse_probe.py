"""The LayerNorm inside the GEMM launch (mmf_gemm_desc.ln_*, gemm_wide.h ln_panel_phase) against GEMM + mmf_layernorm_fwd as two launches: same bits, time per pair.

    python tools/ln_fuse_probe.py [rounds] [iters] [M]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmf_amd import _native as nat


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 7296
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    mk = lambda r, c, s=0.5: (torch.randn(r, c, device=dev, generator=g) * s).bfloat16()
    sync = torch.zeros(nat.GEMM_LN_SYNC_WORDS, dtype=torch.int32, device=dev)
    for name, N, K in (("out_proj", 768, 768), ("ffn_down", 768, 3072)):
        A, W, R = mk(M, K), mk(N, K, 0.03), mk(M, N)
        bias = torch.randn(N, device=dev, generator=g) * 0.1; gamma = torch.rand(N, device=dev, generator=g) + 0.5; beta = torch.randn(N, device=dev, generator=g) * 0.1
        drop = nat.drop_cfg(0.1, 99, None)
        kw = dict(bias=bias, resid=R, ldr=N, drop=drop, debug_flags=nat.gemm_site(nat.SITE_ATTN_OUT_FWD if K == 768 else nat.SITE_FFN_DOWN_FWD))
        y0 = torch.empty(M, N, dtype=torch.bfloat16, device=dev); o0 = torch.empty_like(y0); m0 = torch.empty(M, device=dev); r0 = torch.empty(M, device=dev)
        y1 = torch.empty_like(y0); o1 = torch.empty_like(y0); m1 = torch.empty(M, device=dev); r1 = torch.empty(M, device=dev)

        def two():
            nat.gemm(A, W, y0, M, N, K, K, K, N, **kw)
            nat.layernorm_fwd(y0, gamma, beta, o0, m0, r0, M, N, 1e-12)

        def one():
            nat.gemm(A, W, y1, M, N, K, K, K, N, ln=(gamma, beta, o1, m1, r1, 1e-12, sync), **kw)

        print(name, "M", M, "fusable:", nat.gemm_ln_fusable(A, W, y0, M, N, K, K, K, N, **kw), flush=True)
        two(); kern2 = nat.gemm_last_kernel()
        for rep in range(3):      # the counters carry on from launch to launch
            for t in (y1, o1, m1, r1):
                t.fill_(float("nan"))
            one(); torch.cuda.synchronize()
            print("  ", nat.gemm_last_kernel(), "| two launches:", kern2, "| same bits:", [bool(torch.equal(a, b)) for a, b in ((y0, y1), (o0, o1), (m0, m1), (r0, r1))], flush=True)

        def timed(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters * 1e3

        res = {"two": [], "one": []}
        for _ in range(rounds):
            res["two"].append(timed(two)); res["one"].append(timed(one))
        for k, v in res.items():
            print("   %-4s med %6.1f us  min %6.1f" % (k, statistics.median(v), min(v)), flush=True)
        print("   sync words:", sync[:16].tolist(), flush=True)
        print("   stamps (10 ns): waitcnt+barrier, release, ticket, wait-for-epoch, acquire, claim+rows:", sync.view(-1, 16)[:29, 10:].float().mean(0).tolist(), flush=True)


if __name__ == "__main__":
    main()
