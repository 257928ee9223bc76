"""Where the attention kernels spend their time, per wave: runs the forward, dQ and dK/dV kernels at the VisualBERT VQA2 shape
(B = 32, 12 heads, S = 228, head_dim 64, dropout 0.1, fp32 ctx copy) on an instrumented build of the library and prints the
phase breakdown from the s_memrealtime stamps (100 MHz) every wave records.

    MMF_AMD_EXTRA_HIPCC_FLAGS=-DMMF_ATTN_PROBE python -m mmf_amd.csrc.build --tag probe       # no GPU needed
    MMF_AMD_LIB=mmf_amd/libmmf_amd.probe.so python tools/attn_timeline.py [--nodrop] [--S 228]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmf_amd import _native as nat

PHASES = {
    0: ["issue staging+q", "wait staging", "QK^T", "softmax", "dropout", "PV", "store"],
    1: ["issue staging+delta", "wait staging", "key tile 0", "tiles 1..N/2-1", "tiles N/2..N-1", "store", None],
    2: ["issue staging", "wait staging", "query tile 0", "tiles 1..N/2-1", "tiles N/2..N-1", "store", None],
}
PHASES[3] = ["issue staging+delta", "wait staging", "K fragments", "8-step loop", "store dK, dV", "store dQ", None]
NAMES = {0: "attn_fwd", 1: "attn_bwd_dq", 2: "attn_bwd_dkv", 3: "attn_bwd_fused"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--S", type=int, default=228)
    ap.add_argument("--nodrop", action="store_true")
    ap.add_argument("--noctx32", action="store_true")
    ap.add_argument("--two-pass", action="store_true", help="backward as the separate dQ and dK/dV kernels")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    nat.set_tunable(nat.TUN_ALT_FORMS, 4 * int(args.two_pass))
    B, A, S, D = args.B, 12, args.S, 64
    H = A * D
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = (torch.randn(B * S, 3 * H, device=dev, generator=g) * 0.5).bfloat16()
    mask = torch.zeros(B, S, device=dev)
    ctx = torch.empty(B * S, H, device=dev, dtype=torch.bfloat16)
    ctx32 = None if args.noctx32 else torch.empty(B * S, H, device=dev)
    lse = torch.empty(B * A * S, device=dev)
    dctx = (torch.randn(B * S, H, device=dev, generator=g) * 0.1).bfloat16()
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B * A * S, device=dev)
    drop = nat.NO_DROP if args.nodrop else nat.drop_cfg(0.1, 1234, None)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    dq, dk, dv = dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:]

    def fwd():
        nat.attention_fwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, A, S, S, 0.125, drop=drop, ctx_f32=ctx32)

    def bwd():
        nat.attention_bwd(q, k, v, 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, A, S, S, 0.125, dctx, dq, dk, dv, delta, drop=drop,
                          ctx_f32=ctx32)

    for _ in range(3):
        fwd(); bwd()
    torch.cuda.synchronize()
    cap = 20000
    buf = torch.zeros(12 * (1 + cap), dtype=torch.int64, device=dev)
    nat.attention_set_probe(buf)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record(); fwd(); ev[1].record(); bwd(); ev[2].record()
    torch.cuda.synchronize()
    nat.attention_set_probe(None)
    n = min(int(buf[0].item()), cap)
    rec = buf[12:12 * (1 + n)].view(n, 12).cpu()
    print("shape B=%d heads=%d S=%d d=%d dropout=%s ctx32=%s; events: fwd %.1f us, bwd (dQ + dK/dV) %.1f us (probed build)"
          % (B, A, S, D, not args.nodrop, ctx32 is not None, ev[0].elapsed_time(ev[1]) * 1e3, ev[1].elapsed_time(ev[2]) * 1e3))
    tick = 0.01  # us
    for kid in (0, 1, 2, 3):
        r = rec[rec[:, 0] == kid]
        if r.numel() == 0:
            continue
        t = r[:, 4:].double() * tick
        nstamp = 8 if kid == 0 else 7
        t0 = t[:, 0].min()
        env = (t[:, nstamp - 1].max() - t0).item()
        print("\n%s: %d waves recorded, launch envelope (first entry -> last exit) %.1f us" % (NAMES[kid], r.shape[0], env))
        full = r[:, 2] % 8 < 7          # waves with 32 valid rows (the 8th tile of S = 228 has 4)
        life = t[:, nstamp - 1] - t[:, 0]
        print("  wave lifetime: mean %.2f us, p10 %.2f, p90 %.2f, max %.2f; entry spread: p50 %.2f us, p90 %.2f, max %.2f after the first"
              % (life.mean(), life.quantile(0.1), life.quantile(0.9), life.max(), (t[:, 0] - t0).quantile(0.5), (t[:, 0] - t0).quantile(0.9),
                 (t[:, 0] - t0).max()))
        for i, name in enumerate(PHASES[kid]):
            if name is None or i + 1 >= nstamp:
                continue
            d = t[:, i + 1] - t[:, i]
            print("    %-24s mean %6.2f us   p10 %6.2f   p90 %6.2f" % (name, d.mean(), d.quantile(0.1), d.quantile(0.9)))
        # rounds: waves per (CU) over time — how many waves started in the first 1 us vs later
        started = t[:, 0] - t0
        print("  waves that entered within 2 us of the first: %d of %d; after half the envelope: %d"
              % (int((started < 2).sum()), r.shape[0], int((started > env / 2).sum())))


if __name__ == "__main__":
    main()
