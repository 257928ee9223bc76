"""Micro-benchmark of the fused attention kernels at the VisualBERT VQA2 shape."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmf_amd import _native as nat
B, heads, S = 32, 12, 228
H = heads * 64
dev = "cuda"
qkv = torch.randn(B * S, 3 * H, device=dev).bfloat16()
mask = torch.zeros(B, S, device=dev)
ctx = torch.empty(B * S, H, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, heads, S, device=dev)
dctx = torch.randn(B * S, H, device=dev).bfloat16()
dqkv = torch.empty_like(qkv); delta = torch.empty(B, heads, S, device=dev)
drop = nat.drop_cfg(0.1 if "nodrop" not in sys.argv else 0.0, 1234)
c32 = torch.empty(B * S, H, device=dev) if "noctx32" not in sys.argv else None
if "oldfwd" in sys.argv:
    nat.set_tunable(nat.TUN_ALT_FORMS, 2)
sc = 1 / math.sqrt(64)
def fwd(): nat.attention_fwd(qkv, qkv[:, H:], qkv[:, 2*H:], 3*H, 3*H, 3*H, mask, ctx, H, lse, B, heads, S, S, sc, drop, ctx_f32=c32)
def bwd(): nat.attention_bwd(qkv, qkv[:, H:], qkv[:, 2*H:], 3*H, 3*H, 3*H, mask, ctx, H, lse, B, heads, S, S, sc, dctx, dqkv, dqkv[:, H:], dqkv[:, 2*H:], delta, drop, ctx_f32=c32)
for name, f, fl in (("fwd", fwd, 4), ("bwd", bwd, 14)):
    for _ in range(3): f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%s %.1f us  %.1f TFLOP/s (useful)" % (name, ms * 1e3, fl * B * heads * S * S * 64 / ms / 1e9))
