"""Same-process interleaved A/B of the grouped weight-gradient launch of a VisualBERT layer (dW1, dW2, dWqkv, dWo + their bias
gradients, M = 7296 tokens): 128 x 128 tiles, two workgroups per CU (MMF_TUN_WGRAD_WIDE = 1) against the 256 x 128 wide tile (0).

    python tools/wgrad_ab.py [rounds] [iters] [tokens]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmf_amd import _native as nat


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    T, H, I = (int(sys.argv[3]) if len(sys.argv) > 3 else 7296), 768, 3072      # (T = 6144: the 96 K-steps per CU a stream-K schedule would leave)
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    mk = lambda r, c: (torch.randn(r, c, device=dev, generator=g) * 0.5).bfloat16()
    specs = [(mk(T, I), mk(T, H), I, H), (mk(T, H), mk(T, I), H, I), (mk(T, 3 * H), mk(T, H), 3 * H, H), (mk(T, H), mk(T, H), H, H)]
    probs = []
    for dy, x, N, K in specs:
        dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
        probs.append(dict(A=dy, B=x, C_out=dw, M=N, N=K, K=T, lda=N, ldb=K, ldc=K, a_kmajor=True, b_kmajor=True, rowsum_out=db))
    L = nat.lib()
    flops = sum(2.0 * T * N * K for _, _, N, K in specs)
    vals = [1, 0]
    times = {v: [] for v in vals}
    names = {}
    for r in range(rounds):
        for v in vals:
            L.mmf_amd_set_tunable(nat.TUN_WGRAD_WIDE, v)
            for _ in range(3):
                nat.gemm_grouped(probs)
            names[v] = nat.gemm_last_kernel()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                nat.gemm_grouped(probs)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / iters * 1e3)
    for v in vals:
        med = statistics.median(times[v])
        print("wgrad_wide=%d %-34s med %6.1f us min %6.1f  (%4.0f TFLOP/s)" % (v, names[v], med, min(times[v]), flops / med / 1e6), flush=True)
    L.mmf_amd_set_tunable(nat.TUN_WGRAD_WIDE, 0)


if __name__ == "__main__":
    main()
