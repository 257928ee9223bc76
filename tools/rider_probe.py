"""Feasibility probe for a LayerNorm-backward RIDER on the grouped weight-gradient launch: the layer's weight gradients (216 tiles of 256 x 128 on 256 CUs,
one workgroup per CU, 147 KB of LDS each) and the NEXT layer's first LayerNorm backward (independent of them, HBM-bound, 45 MB) launched on two HIP
streams at once — the LayerNorm's small workgroups can only land on the 40 CUs the tiles leave free — against the two launched back to back.

    python tools/rider_probe.py [rounds] [iters] [layernorm rows]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmf_amd import _native as nat


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    T, H, I = 7296, 768, 3072
    TL = int(sys.argv[3]) if len(sys.argv) > 3 else T       # rows of the LayerNorm (a multiple of T: how much streaming work the 40 idle CUs absorb before they outlast the tiles)
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    mk = lambda r, c: (torch.randn(r, c, device=dev, generator=g) * 0.5).bfloat16()
    specs = [(mk(T, I), mk(T, H), I, H), (mk(T, H), mk(T, I), H, I), (mk(T, 3 * H), mk(T, H), 3 * H, H), (mk(T, H), mk(T, H), H, H)]
    probs = []
    for dy, x, N, K in specs:
        dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
        probs.append(dict(A=dy, B=x, C_out=dw, M=N, N=K, K=T, lda=N, ldb=K, ldc=K, a_kmajor=True, b_kmajor=True, rowsum_out=db))
    dy, x = mk(TL, H), mk(TL, H)
    mean = torch.zeros(TL, device=dev); rstd = torch.ones(TL, device=dev); gamma = torch.ones(H, device=dev)
    dx = torch.empty_like(x); dlin = torch.empty_like(x)
    ws = torch.zeros(nat.layernorm_bwd_ws_floats(H), device=dev)
    drop = nat.drop_cfg(0.1, 77, None)

    def ln():
        nat.layernorm_bwd(dy, x, mean, rstd, gamma, dx, dlin, drop, None, None, None, 0, ws, TL, H)

    def wg():
        nat.gemm_grouped(probs)

    def rider():
        nat.gemm_grouped_ln(probs, dy, x, mean, rstd, gamma, dx, dlin, drop, ws, TL, H)

    # same bits: the two launches one after the other against the one launch with the rider
    both = lambda: (wg(), ln())
    both(); torch.cuda.synchronize()
    want = [p["C_out"].clone() for p in probs] + [p["rowsum_out"].clone() for p in probs] + [dx.clone(), dlin.clone(), ws.clone()]
    for t in [p["C_out"] for p in probs] + [p["rowsum_out"] for p in probs] + [dx, dlin, ws]:
        t.zero_()
    rider(); torch.cuda.synchronize()
    got = [p["C_out"] for p in probs] + [p["rowsum_out"] for p in probs] + [dx, dlin, ws]
    print("rider launched:", nat.gemm_last_kernel(), " bit-identical:", [bool(torch.equal(a, b)) for a, b in zip(want, got)], flush=True)

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

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

    def both_serial():
        wg(); ln()

    def both_parallel():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            wg()
        with torch.cuda.stream(s2):
            ln()
        cur.wait_stream(s1); cur.wait_stream(s2)

    res = {k: [] for k in ("wgrad", "ln_bwd", "serial", "parallel", "rider")}
    for _ in range(rounds):
        res["wgrad"].append(timed(wg)); res["ln_bwd"].append(timed(ln)); res["serial"].append(timed(both_serial)); res["parallel"].append(timed(both_parallel)); res["rider"].append(timed(rider))
    for k, v in res.items():
        print("%-9s med %6.1f us  min %6.1f" % (k, statistics.median(v), min(v)), flush=True)


if __name__ == "__main__":
    main()
