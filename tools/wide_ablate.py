"""(needs a library built with MMF_AMD_EXTRA_HIPCC_FLAGS=-DMMF_WIDE_ABLATE python -m mmf_amd.csrc.build --force)
K-loop anatomy of the wide-tile GEMM by ablation (debug_flags bits 4-6: no DMA issue in the loop / no MFMA / no fragment
reads; results are garbage, only the probe's K-loop time means something) at the encoder's forward shapes, hot operands."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmf_amd import _native as nat

VARIANTS = (("full", 0), ("no-dma", 1), ("no-mfma", 2), ("no-reads", 4), ("dma-only", 6), ("reads-only", 3), ("mfma-only", 5), ("barriers-only", 7))


def kloop_us(f, reps=5):
    buf = torch.zeros(8 * (1 + 4096), dtype=torch.int64, device="cuda")
    out = []
    for _ in range(reps):
        buf.zero_()
        nat.gemm_set_probe(buf)
        f()
        torch.cuda.synchronize()
        nat.gemm_set_probe(None)
        n = int(buf[0].item())
        r = buf[8:8 * (1 + n)].view(n, 8).double()
        out.append(((r[:, 4] - r[:, 3]).mean().item() * 0.01, (r[:, 3] - r[:, 2]).mean().item() * 0.01, (r[:, 6] - r[:, 4]).mean().item() * 0.01,
                    (r[:, 6].max() - r[:, 2].min()).item() * 0.01))
    out.sort()
    return out[len(out) // 2]


def main():
    M = 7296
    for N, K, cfg in ((768, 768, 1), (2304, 768, 2), (3072, 768, 3), (768, 3072, 1), (3072, 768, 2)):
        A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        nat.set_tunable(nat.TUN_GEMM_WIDE, cfg)
        line = "N=%4d K=%4d cfg %d: " % (N, K, cfg)
        for name, bits in VARIANTS:
            kl, pro, epi, env = kloop_us(lambda: nat.gemm(A, B, C, M, N, K, K, K, N, debug_flags=bits << 4))
            line += " %s %.2f/step" % (name, kl / (K // 64))
            if name == "full":
                line += " (prolog %.1f, epilogue %.1f, envelope %.1f us)" % (pro, epi, env)
        print(line, flush=True)
        nat.set_tunable(nat.TUN_GEMM_WIDE, -1)
        kl, pro, epi, env = kloop_us(lambda: nat.gemm(A, B, C, M, N, K, K, K, N))
        print("      128-row kernel: k-loop %.2f/step (prolog %.1f, epilogue %.1f, envelope %.1f us)" % (kl / (K // 64), pro, epi, env), flush=True)
    nat.set_tunable(nat.TUN_GEMM_WIDE, 0)


if __name__ == "__main__":
    main()
