"""Throughput of the fp32-accurate forward path (mmf_amd.fp32_inference(), mmf_amd/csrc/fp32_path.hip) on one MI355X, beside
the bf16 eval forward of the same model and the vendor fp32 GEMM (torch.matmul on fp32 operands -> hipBLASLt / rocBLAS, the
"reference PyTorch path on this GPU" for these contractions).

  * per GEMM shape of a VisualBERT VQA2 encoder layer (M = 7296): mmf_gemm_f32 µs / TFLOP/s against the 157.3 TFLOP/s fp32-input
    MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md), torch.matmul fp32 beside it;
  * attention_f32_fwd and layernorm_f32_fwd per launch;
  * the whole eval forward at B = 32 in fp32 and in bf16 (ms, samples/s).

Writes gpurun_out/fp32_bench.json."""
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mmf_amd
from mmf_amd import _native as nat
from bench import build, synthetic_batch

PEAK_F32_MFMA = 157.3


def timed(f, iters=10, warm=2):
    for _ in range(warm):
        f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = "cuda"
    res = {"peak_tflops_f32_mfma": PEAK_F32_MFMA, "gemm": [], "note": "fp32 operands and outputs, epilogue = bias only"}
    M = 7296
    for name, N, K in (("qkv (packed)", 2304, 768), ("out-proj", 768, 768), ("ffn-up", 3072, 768), ("ffn-down", 768, 3072),
                       ("visual projection", 768, 2048)):
        m = 3200 if name.startswith("visual") else M
        A = torch.randn(m, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
        C = torch.empty(m, N, device=dev)
        t = timed(lambda: nat.gemm_f32(A, W, C, m, N, K, K, K, N, bias=b))
        tl = timed(lambda: torch.addmm(b, A, W.t(), out=C))
        fl = 2.0 * m * N * K
        res["gemm"].append(dict(name=name, M=m, N=N, K=K, us=round(t * 1e3, 1), tflops=round(fl / t / 1e9, 1),
                                frac_of_peak=round(fl / t / 1e9 / PEAK_F32_MFMA, 3), torch_fp32_us=round(tl * 1e3, 1),
                                torch_fp32_tflops=round(fl / tl / 1e9, 1)))
        print(res["gemm"][-1], flush=True)
    B, heads, S, H = 32, 12, 228, 768
    qkv = torch.randn(B * S, 3 * H, device=dev); ctx = torch.empty(B * S, H, device=dev); mask = torch.zeros(B, S, device=dev)
    t = timed(lambda: nat.attention_f32_fwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, mask, ctx, H, B, heads, S, S, 0.125))
    res["attention_f32_fwd"] = dict(us=round(t * 1e3, 1), tflops=round(4.0 * B * heads * S * S * 64 / t / 1e9, 1))
    # ViLBERT (BASELINE configs[3]): visual stream 8 heads x 128 over 101 regions; co-attention 128 text queries over 101 region keys
    Bv, hv, R, T, Hv = 32, 8, 101, 128, 1024
    qv = torch.randn(Bv * R, 3 * Hv, device=dev); qt = torch.randn(Bv * T, 3 * Hv, device=dev)
    cv = torch.empty(Bv * R, Hv, device=dev); ct = torch.empty(Bv * T, Hv, device=dev)
    sc = 1.0 / math.sqrt(128.0)
    t = timed(lambda: nat.attention_f32_fwd(qv, qv[:, Hv:], qv[:, 2 * Hv:], 3 * Hv, 3 * Hv, 3 * Hv, None, cv, Hv, Bv, hv, R, R, sc, head_dim=128))
    res["attention_f32_fwd_d128_self_101"] = dict(us=round(t * 1e3, 1), tflops=round(4.0 * Bv * hv * R * R * 128 / t / 1e9, 1))
    t = timed(lambda: nat.attention_f32_fwd(qt, qv[:, Hv:], qv[:, 2 * Hv:], 3 * Hv, 3 * Hv, 3 * Hv, None, ct, Hv, Bv, hv, T, R, sc, head_dim=128))
    res["attention_f32_fwd_d128_cross_128x101"] = dict(us=round(t * 1e3, 1), tflops=round(4.0 * Bv * hv * T * R * 128 / t / 1e9, 1))
    print(res["attention_f32_fwd_d128_self_101"], res["attention_f32_fwd_d128_cross_128x101"], flush=True)
    x = torch.randn(B * S, H, device=dev); g = torch.ones(H, device=dev); be = torch.zeros(H, device=dev); y = torch.empty_like(x)
    t = timed(lambda: nat.layernorm_f32_fwd(x, g, be, y, B * S, H, 1e-12))
    res["layernorm_f32_fwd"] = dict(us=round(t * 1e3, 1), gbps=round(2.0 * x.numel() * 4 / t / 1e6, 1))
    print(res["attention_f32_fwd"], res["layernorm_f32_fwd"], flush=True)

    model = build(dev, 0); model.eval()          # bench.py's model and batch: VisualBERT-base VQA2, B = 32, 128 tokens + 100 x 2048
    batch = synthetic_batch(32, 0, dev)

    def fwd32():
        with mmf_amd.fp32_inference():
            return model(batch)["scores"]

    def fwd16():
        with torch.no_grad():
            return model(batch)["scores"]

    # attention backward (dQ launch + dK / dV launch) at the VisualBERT shape
    dq = torch.empty_like(qkv); dctx = torch.randn(B * S, H, device=dev); lse = torch.empty(B, heads, S, device=dev); dl = torch.empty(B, heads, S, device=dev)
    nat.attention_f32_fwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, mask, ctx, H, B, heads, S, S, 0.125, lse=lse)
    t = timed(lambda: nat.attention_f32_bwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, mask, ctx, H, lse, B, heads, S, S, 0.125, dctx,
                                            dq, dq[:, H:], dq[:, 2 * H:], dl))
    res["attention_f32_bwd"] = dict(us=round(t * 1e3, 1), tflops=round(14.0 * B * heads * S * S * 64 / t / 1e9, 1),
                                    note="7 S x S x 64 products per (batch, head): S^T, dP^T, dQ in the first launch; S, dP, dV, dK in the second")
    # weight-gradient layout (split-K) and dgrad layout of the FFN shapes
    dy = torch.randn(M, 3072, device=dev); xx = torch.randn(M, 768, device=dev); dw = torch.empty(3072, 768, device=dev)
    t = timed(lambda: nat.gemm_f32(dy, xx, dw, 3072, 768, M, 3072, 768, 768, a_kmajor=True, b_kmajor=True, split_k=True))
    res["gemm_wgrad_3072x768_K7296"] = dict(us=round(t * 1e3, 1), tflops=round(2.0 * M * 3072 * 768 / t / 1e9, 1))
    W = torch.randn(3072, 768, device=dev); dx = torch.empty(M, 768, device=dev)
    t = timed(lambda: nat.gemm_f32(dy, W, dx, M, 768, 3072, 3072, 768, 768, b_kmajor=True))
    res["gemm_dgrad_768_K3072"] = dict(us=round(t * 1e3, 1), tflops=round(2.0 * M * 3072 * 768 / t / 1e9, 1))
    print(res["attention_f32_bwd"], res["gemm_wgrad_3072x768_K7296"], res["gemm_dgrad_768_K3072"], flush=True)

    t32, t16 = timed(fwd32, iters=5), timed(fwd16, iters=5)
    flops = 32 * 40.97e9
    res["eval_forward_B32"] = dict(fp32_ms=round(t32, 2), fp32_samples_per_s=round(32e3 / t32, 1), fp32_tflops=round(flops / t32 / 1e9, 1),
                                   bf16_ms=round(t16, 2), bf16_samples_per_s=round(32e3 / t16, 1),
                                   max_abs_score_diff_bf16_vs_fp32=float((fwd32() - fwd16().float()).abs().max()))
    print(res["eval_forward_B32"], flush=True)
    # the fp32 TRAINING step (mmf_amd.fp32_training(): forward + logit_bce + backward on the fp32 kernels, fused AdamW), train mode, B = 32
    from mmf_amd.modules.optimizers import AdamW
    model.train()
    opt = AdamW(model.parameters(), lr=5e-5, weight_decay=0.01)

    def train32():
        opt.zero_grad()
        with mmf_amd.fp32_training():
            out = model(batch)
        list(out["losses"].values())[0].backward()
        opt.step()

    tt = timed(train32, iters=4, warm=2)
    res["train_step_fp32_B32"] = dict(ms=round(tt, 2), samples_per_s=round(32e3 / tt, 1), tflops=round(32 * 122.9e9 / tt / 1e9, 1),
                                      frac_of_fp32_mfma_peak=round(32 * 122.9e9 / tt / 1e9 / PEAK_F32_MFMA, 3))
    print(res["train_step_fp32_B32"], flush=True)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(res, open(os.path.join(out, "fp32_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
