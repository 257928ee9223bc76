"""Where does a layer GEMM's "in-situ penalty" come from?  (DESIGN round-3 section 7 item 0: the same kernels run 15 - 30 % slower inside
the training step than on warm operands in isolation.)

    python tools/cold_operand_probe.py [shape-name-filter ...]

One consumer GEMM (a layer shape with its real epilogue) is timed per launch (HIP events around that launch only) while the state of its
A operand is varied:

    warm      the same A every launch (what tools/gemm_ab.py measures: A sits in the Infinity Cache / L2)
    rotate    8 different A buffers in turn (8 x 45 MB > the 256 MB Infinity Cache for the K = 3072 shapes): A comes from HBM
    fresh     A was written by an elementwise kernel (plain stores, 16 B per lane) immediately before the launch
    fresh-nt  the same through this library's GEMM epilogue (the in-step producer: FFN-up's GELU epilogue, non-temporal stores)
    fresh-gemm-t / -ct   that producer with temporal stores (MMF_TUN_EPI_NT = 1: every output; = 3: the bf16 output only, gelu' still nt)
    hot       warm A, but a heavy unrelated GEMM runs immediately before (the chip at the step's power / clock state)
    side-rotate / side-A-rotate   the epilogue's row-wise side input (residual / saved gelu') from 8 buffers in turn (cold), A warm / cold too
    out-sc1   warm operands, the consumer's own output stored write-through instead of non-temporally
    l2flush   64 MB of unrelated data written before every launch: the L2s lose the operands, the Infinity Cache keeps them
    flush     600 MB of unrelated data written before every launch: A, the side input AND the weights come from HBM (the in-step state of a
              weight panel: 283 MB of bf16 weights and twins, 1.4 GB of activation traffic per step)

`fresh` ~ `warm` would say the freshly written activation is served from cache and the penalty is clocks; `fresh` ~ `rotate` says the
consumer pulls it from HBM whatever the producer's store policy."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmf_amd import _native as nat
from tools.gemm_ab import SHAPES, make, M


def timed_launches(fn_before, fn, n):
    ts = []
    for _ in range(n):
        if fn_before is not None:
            fn_before()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 for a, b in ts]


def main():
    filt = sys.argv[1:]
    dev = "cuda"
    # an unrelated heavy GEMM (the `hot` mode's neighbour) and the in-step producer of the K = 3072 operands (FFN-up + GELU)
    Xh = torch.randn(M, 768, device=dev).bfloat16(); Wh = (torch.randn(3072, 768, device=dev) * 0.05).bfloat16()
    Ch = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16); Uh = torch.empty_like(Ch)
    bias_h = torch.zeros(3072, device=dev)
    big = torch.zeros(150 * 1024 * 1024, device=dev)
    mid = torch.zeros(16 * 1024 * 1024, device=dev)
    for name, N, K, kind in SHAPES:
        if filt and not any(f in name for f in filt):
            continue
        A, B, C, kw, tkw, ref, K, N = make(name, N, K, kind)
        rot = [A] + [A.clone() for _ in range(7)]
        src = A.clone()
        run = lambda a=A: nat.gemm(a, B, C, M, N, K, K, K, N, **tkw)
        res = {}
        for rnd in range(5):
            for mode in ("warm", "rotate", "fresh", "fresh-nt", "fresh-gemm-t", "fresh-gemm-ct", "fresh-gemm-sc1", "fresh-gemm-csc1", "hot", "side-rotate", "side-A-rotate", "out-sc1", "flush", "l2flush"):
                if mode == "warm":
                    run(); t = timed_launches(None, run, 16)
                elif mode == "rotate":
                    idx = [0]
                    def f():
                        idx[0] = (idx[0] + 1) % 8
                        nat.gemm(rot[idx[0]], B, C, M, N, K, K, K, N, **tkw)
                    t = timed_launches(None, f, 16)
                elif mode == "fresh":
                    t = timed_launches(lambda: torch.add(src, 0.0, out=A), run, 16)
                elif mode == "fresh-nt":
                    if K != 3072:
                        continue
                    # FFN-up writes its [M, 3072] output straight into A (same bytes, the real producer's store pattern and policy)
                    t = timed_launches(lambda: nat.gemm(Xh, Wh, A, M, 3072, 768, 768, 768, 3072, bias=bias_h, act=1, U=Uh), run, 16)
                elif mode.startswith("fresh-gemm-"):
                    if K != 3072:
                        continue
                    # the same producer with other store policies (MMF_TUN_EPI_NT, MMF_TUN_EPI_SC1): t every output temporal; ct the bf16 output C
                    # temporal, the saved gelu' still nt; sc1 every output write-through (no nt); csc1 C write-through, gelu' nt
                    pol, sc1 = {"fresh-gemm-t": (1, 0), "fresh-gemm-ct": (3, 0), "fresh-gemm-sc1": (1, 7), "fresh-gemm-csc1": (3, 1)}[mode]
                    def prod():
                        nat.set_tunable(6, pol); nat.set_tunable(12, sc1)
                        nat.gemm(Xh, Wh, A, M, 3072, 768, 768, 768, 3072, bias=bias_h, act=1, U=Uh)
                        nat.set_tunable(6, 0); nat.set_tunable(12, 0)
                    t = timed_launches(prod, run, 16)
                elif mode in ("side-rotate", "side-A-rotate"):
                    # the epilogue's row-wise side input (residual rows / the act-2 multiplier = the saved gelu') from HBM: 8 buffers in turn
                    key = "aux" if "aux" in tkw else ("resid" if "resid" in tkw else None)
                    if key is None:
                        continue
                    if "side_rot" not in res:
                        res["side_rot"] = None
                        side_rot = [tkw[key]] + [tkw[key].clone() for _ in range(7)]
                    idx = [0]
                    def f2():
                        idx[0] = (idx[0] + 1) % 8
                        kw2 = dict(tkw); kw2[key] = side_rot[idx[0]]
                        nat.gemm(rot[idx[0]] if mode == "side-A-rotate" else A, B, C, M, N, K, K, K, N, **kw2)
                    t = timed_launches(None, f2, 16)
                elif mode == "flush":
                    # everything cold, the weights too: 600 MB of other data go through the caches before every launch
                    t = timed_launches(lambda: big.add_(1.0), run, 16)
                elif mode == "l2flush":
                    # 64 MB of unrelated data in between: every XCD's 4 MB L2 is refilled, the 256 MB Infinity Cache still holds the operands
                    t = timed_launches(lambda: mid.add_(1.0), run, 16)
                elif mode == "out-sc1":
                    def f3():
                        nat.set_tunable(6, 1); nat.set_tunable(12, 7)
                        run()
                        nat.set_tunable(6, 0); nat.set_tunable(12, 0)
                    t = timed_launches(None, f3, 16)
                else:
                    t = timed_launches(lambda: nat.gemm(Xh, Wh, Ch, M, 3072, 768, 768, 768, 3072, bias=bias_h, act=1, U=Uh), run, 16)
                res.setdefault(mode, []).append(statistics.median(t))
        fl = 2.0 * M * N * K
        print("%-11s N=%4d K=%4d %-24s" % (name, N, K, nat.gemm_last_kernel()[-24:]) +
              "  ".join("%s %5.1f us (%3.0f TF)" % (m, statistics.median(v), fl / statistics.median(v) / 1e6) for m, v in res.items() if v), flush=True)


if __name__ == "__main__":
    main()
