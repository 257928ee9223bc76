"""Inference-only (no autograd) row-wise pass of a `BertLayerJit` over a per-sample K|V cache — the building block of M4C's
incremental greedy decoding.

The reference decodes by re-running the whole multimodal transformer `dec_step_num` times (mmf/models/m4c.py:284-304); with
its prefix-LM mask (:424-440) the encoder positions never see a decoding position, so their hidden states — and their keys and
values in every layer — are the same in all steps, and decoding row t only ever sees rows <= t.  `layer_rows` therefore
projects ONLY the new rows, writes their Q|K|V into the cache rows they own (GEMM epilogue row remap, no copy), attends from
them to the `Sk` cache rows that exist so far, and finishes the layer (output projection + residual + LayerNorm, GELU
feed-forward + residual + LayerNorm) on those rows alone.  Same kernels as the training path, dropout off.
"""
import math

import torch

from mmf_amd import functional as Fn

nat = Fn.nat
BF16, F32 = torch.bfloat16, torch.float32


def layer_rows(layer, x2, cache, B, rows_per_sample, cache_rows, row0, Sk, key_mask_add):
    """x2: bf16 [B * rows_per_sample, H] — the new rows (rows_per_sample per sample, consecutive positions row0 ...).
    cache: bf16 [B * cache_rows, 3H] — this layer's packed Q|K|V for every position of every sample.
    Sk: keys visible to the new rows = cache rows [0, Sk) of their sample (Sk >= row0 + rows_per_sample).
    key_mask_add: fp32 [B, cache_rows] additive key mask (0 / -10000) or None.  Returns the layer output [B * rows, H]."""
    sa, so = layer.attention.self, layer.attention.output
    it, ot = layer.intermediate, layer.output
    H = x2.shape[1]
    heads = sa.num_attention_heads
    R = rows_per_sample
    M = B * R
    dev = x2.device
    w16, b32 = sa.packed_qkv()
    # Q|K|V of the new rows straight into their cache rows: row = m + (m / R) * (cache_rows - R) + row0
    nat.gemm(x2, w16, cache, M, 3 * H, H, H, H, 3 * H, bias=b32, grp=(R, cache_rows - R, row0))
    ctxt = torch.empty(M, H, dtype=BF16, device=dev)
    lse = torch.empty(B, heads, R, dtype=F32, device=dev)
    q = cache[row0:]
    nat.attention_fwd(q, cache[:, H:], cache[:, 2 * H:], 3 * H, 3 * H, 3 * H, key_mask_add, ctxt, H, lse, B, heads, R, Sk,
                      1.0 / math.sqrt(H // heads), head_dim=H // heads, q_batch_rows=cache_rows, kv_batch_rows=cache_rows,
                      mask_batch_stride=cache_rows if key_mask_add is not None else 0)
    y = torch.empty(M, H, dtype=BF16, device=dev)
    nat.gemm(ctxt, Fn.shadows.get(so.dense.weight), y, M, H, H, H, H, H, bias=so.dense.bias.detach(), resid=x2, ldr=H)
    a_out = torch.empty(M, H, dtype=BF16, device=dev)
    nat.layernorm_fwd(y, so.LayerNorm.weight.detach(), so.LayerNorm.bias.detach(), a_out, None, None, M, H, so.LayerNorm.eps)
    I = it.dense.weight.shape[0]
    hh = torch.empty(M, I, dtype=BF16, device=dev)
    nat.gemm(a_out, Fn.shadows.get(it.dense.weight), hh, M, I, H, H, H, I, bias=it.dense.bias.detach(), act=1)
    y2 = torch.empty(M, H, dtype=BF16, device=dev)
    # (one launch: the skinny split-K path — a workspace allocation and a second kernel per call — is for the training heads' long
    # reductions, not for a host-bound decoding loop; bit 18 of debug_flags keeps this K = 3072 GEMM on the single-kernel path)
    nat.gemm(hh, Fn.shadows.get(ot.dense.weight), y2, M, H, I, I, I, H, bias=ot.dense.bias.detach(), resid=a_out, ldr=H,
             debug_flags=nat.GEMM_NO_SKINNY)
    out = torch.empty(M, H, dtype=BF16, device=dev)
    nat.layernorm_fwd(y2, ot.LayerNorm.weight.detach(), ot.LayerNorm.bias.detach(), out, None, None, M, H, ot.LayerNorm.eps)
    return out


def new_cache(B, cache_rows, H, layers, device):
    """One packed Q|K|V buffer per layer, zero-initialised (rows that no step has written yet are never addressed as keys)."""
    return [torch.zeros(B * cache_rows, 3 * H, dtype=BF16, device=device) for _ in range(layers)]
