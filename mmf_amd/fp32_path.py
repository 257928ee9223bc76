"""The fp32-accurate forward path: the same operators, activations kept in fp32, contractions on the fp32-input matrix cores.

The reference computes in fp32 unless `training.fp16` turns autocast on (mmf/trainers/core/training_loop.py:199), and
BASELINE.json's north_star states two bounds: 5e-2 for bf16 and 1e-3 for fp32.  The throughput path (mmf_amd/functional.py) is
bf16 and meets the first; inside

    with mmf_amd.fp32_inference():
        out = model(sample_list)

every `torch.ops.mmf_amd.*` operator routes to the kernels of mmf_amd/csrc/fp32_path.hip instead (`mmf_gemm_f32` on
`v_mfma_f32_32x32x2_f32` — exact fp32 products, fp32 accumulation — `mmf_attention_f32_fwd`, `mmf_layernorm_f32_fwd`, fp32
embedding gathers), reading the fp32 master parameters directly (no bf16 shadows).  The context implies `torch.no_grad()`: it is
an evaluation / inference / parity-checking mode (forward only), and a module in training mode with a non-zero dropout
probability is an error, not a silent no-op.  Same models, same parameter names, same `forward(sample_list)`.

Reference operations, as in mmf_amd/functional.py: BertVisioLinguisticEmbeddings.forward (mmf/modules/embeddings.py:423-459),
BertLayerJit.forward (mmf/modules/hf_layers.py:255-292), BertPooler / BertPredictionHeadTransform / classifier Linear
(mmf/models/visual_bert.py:146,327-330,389-401).
"""
import contextlib
import math
import weakref

import torch

from mmf_amd import _native as nat

F32 = torch.float32
_depth = 0


def active():
    return _depth > 0


@contextlib.contextmanager
def fp32_inference():
    """Run every mmf_amd operator inside the block on the fp32 kernels (forward only, gradients off)."""
    global _depth
    from mmf_amd import _ops_native
    _depth += 1
    _ops_native.push_mode(0)        # the native operators forward to the fp32 kernels' Python bindings while this is on
    try:
        with torch.no_grad():
            yield
    finally:
        _depth -= 1
        _ops_native.pop_mode(0)


def check_no_dropout(p, training):
    if training and p is not None and p > 0.0:
        raise RuntimeError("mmf_amd.fp32_inference() is a forward-only evaluation mode: call model.eval() first "
                           "(a dropout site with p = %g is in training mode)" % p)


def _rows(x):
    """fp32, contiguous, token-major [rows, features] view of an activation or input."""
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype != F32:
        raise TypeError("fp32 path: expected a float32 tensor, got %s (activations born outside fp32_inference()?)" % x2.dtype)
    return x2 if x2.is_contiguous() else x2.contiguous()


def _w(p):
    w = p.detach()
    return w if w.is_contiguous() else w.contiguous()


_packs = {}


def _packed(*params):
    """One contiguous fp32 buffer holding `params` stacked along dim 0 (the Q | K | V weights as one [3H, H] operand, their biases as
    one [3H] vector), rebuilt when a parameter's version or storage changes.  A cache of its own: the bf16 shadows of
    functional.ShadowCache are mirrors the fused optimizer writes to, these are plain copies the forward-only path reads."""
    head = params[0]
    key = id(head)
    sig = tuple((id(p), p._version, p.data_ptr()) for p in params)
    ent = _packs.get(key)
    if ent is not None and ent[0] == sig:
        return ent[1]
    if ent is None:
        weakref.finalize(head, _packs.pop, key, None)
    rows = sum(p.shape[0] for p in params)
    buf = ent[1] if ent is not None and ent[1].shape[0] == rows else torch.empty((rows,) + tuple(head.shape[1:]), dtype=F32, device=head.device)
    r = 0
    for p in params:
        buf[r:r + p.shape[0]].copy_(p.detach())
        r += p.shape[0]
    _packs[key] = (sig, buf)
    return buf


def _linear(x2, weight, bias, out=None, ldc=None, act=0, resid=None, **kw):
    M, K = x2.shape
    N = weight.shape[0]
    if K % 4:
        raise ValueError("fp32 path: the contraction length (%d) must be a multiple of 4" % K)
    if out is None:
        out = torch.empty(M, N, dtype=F32, device=x2.device)
        ldc = N
    nat.gemm_f32(x2, _w(weight), out, M, N, K, K, K, ldc, bias=None if bias is None else _w(bias), act=act, resid=resid,
                 ldr=0 if resid is None else resid.stride(0), **kw)
    return out


def layer_norm(x, gamma, beta, eps):
    x2 = _rows(x)
    M, H = x2.shape
    y = torch.empty(M, H, dtype=F32, device=x2.device)
    nat.layernorm_f32_fwd(x2, _w(gamma), _w(beta), y, M, H, eps)
    return y.view(x.shape)


def linear(x, weight, bias):
    return _linear(_rows(x), weight, bias).view(*x.shape[:-1], weight.shape[0])


def dense_gelu(x, weight, bias):
    return _linear(_rows(x), weight, bias, act=1).view(*x.shape[:-1], weight.shape[0])


def linear_tanh(x, weight, bias):
    return _linear(_rows(x), weight, bias, act=3).view(*x.shape[:-1], weight.shape[0])


def gather_rows(x, index):
    B, S, H = x.shape
    out = torch.empty(B, H, dtype=F32, device=x.device)
    nat.gather_rows_f32(_rows(x), index.contiguous().long(), out, B, S, H)
    return out


def pair_halves(x):
    """nlvr2 pairing [2B, H] -> [B, 2H] = cat(x[:B], x[B:], dim=1) (visual_bert.py:369-374): two strided row copies by the row-copy
    kernel (fp32 rows moved as pairs of 16-bit words)."""
    x2 = _rows(x)
    B2, H = x2.shape
    B = B2 // 2
    out = torch.empty(B, 2 * H, dtype=F32, device=x2.device)
    xb = x2.view(torch.bfloat16)                              # [2B, 2H] 16-bit words
    ob = out.view(torch.bfloat16).view(2 * B, 2 * H)
    nat.copy_rows(xb, 1, ob, 2, B, 1, 2 * H)                  # first image  -> columns [0, H)
    nat.copy_rows(xb[B:], 1, ob[1:], 2, B, 1, 2 * H)          # second image -> columns [H, 2H)
    return out


def visio_linguistic_embeddings(input_ids, token_type_ids, feats, vtype, word, pos, typ, ln_w, ln_b, typ_vis, pos_vis, proj_w, proj_b,
                                eps, align=None):
    """embeddings.py:423-459 with image_text_alignment=None: text rows = word + position + type; visual rows = projection(features)
    + visual type + visual position 0, written by the projection GEMM's epilogue into rows T.. of the joint sequence; LayerNorm."""
    B, T = input_ids.shape
    H = word.shape[1]
    R = 0 if feats is None else feats.shape[1]
    S = T + R
    dev = word.device
    y = torch.empty(B * S, H, dtype=F32, device=dev)
    nat.embed_text_f32_fwd(input_ids.contiguous(), token_type_ids.contiguous(), _w(word), _w(pos), _w(typ), y, B, T, S, H)
    if R:
        D = feats.shape[2]
        f2 = feats.reshape(B * R, D)
        f2 = (f2 if f2.dtype == F32 else f2.float()).contiguous()
        if D % 4:
            raise ValueError("fp32 path: visual feature width (%d) must be a multiple of 4" % D)
        vt = vtype.reshape(B * R).contiguous()
        if align is not None:        # image_text_alignment (embeddings.py:373-397): per region, the mean text-position row of its aligned words + visual type
            al = align.reshape(B * R, -1).long().contiguous()
            addend = torch.empty(B * R, H, dtype=F32, device=dev)
            nat.align_pos_fwd(al, _w(pos), _w(typ_vis), vt, addend, B * R, al.shape[1], H)
            nat.gemm_f32(f2, _w(proj_w), y, B * R, H, D, D, D, H, bias=_w(proj_b), coladd=_w(pos_vis)[0], rowtab=addend,
                         rowidx=torch.arange(B * R, dtype=torch.int64, device=dev), rowtab_ld=H, grp=(R, T, T))
        else:
            nat.gemm_f32(f2, _w(proj_w), y, B * R, H, D, D, D, H, bias=_w(proj_b), coladd=_w(pos_vis)[0], rowtab=_w(typ_vis),
                         rowidx=vt, rowtab_ld=H, grp=(R, T, T))
    out = torch.empty(B * S, H, dtype=F32, device=dev)
    nat.layernorm_f32_fwd(y, _w(ln_w), _w(ln_b), out, B * S, H, eps)
    return out.view(B, S, H)


def _check_head(hd, Sk):
    if hd not in (64, 128):
        raise NotImplementedError("fp32 path: the attention kernel is built for head_dim 64 and 128, got %d" % hd)
    if Sk > (512 if hd == 64 else 256):      # (beyond 256 / 128 keys the kernels stage K / V in blocks; the cap is the bf16 kernels' own)
        raise NotImplementedError("fp32 path: %d positions exceed what the attention kernels take (512 at head_dim 64, 256 at head_dim 128)" % Sk)


def _attn_mask(mask_add, B, S):
    """The additive mask in the form the fp32 attention takes it: the key mask [B, S], or — a materialised [B, 1, S, S] mask, hf_layers.py:187-190 —
    one row per query [B, S, S] (mmf_attn_desc.mask_query_stride)."""
    if mask_add is None:
        return None
    m = mask_add.float()
    if m.numel() == B * S:
        return m.reshape(B, S).contiguous()
    if m.numel() == B * S * S:
        return m.reshape(B, S, S).contiguous()
    return m.reshape(B, -1, S, S).contiguous()      # one [S, S] mask per head, [B, heads, S, S] (mmf_attn_desc.mask_head_stride)


def transformer_layer(x, wq, bq, wk, bk, wv, bv, wo, bo, ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b, mask_add, heads, eps1, eps2,
                      causal_tail=0, qk_gate=None):
    """BertLayerJit.forward (hf_layers.py:255-292), eval mode: the packed Q|K|V projection into one [M, 3H] buffer, fused attention,
    output projection + bias + residual in the GEMM epilogue, LayerNorm, GELU in the up-projection epilogue, down-projection +
    bias + residual, LayerNorm."""
    B, S, H = x.shape
    hd = H // heads
    _check_head(hd, S)
    x2 = _rows(x)
    M = B * S
    dev = x2.device
    qkv = _linear(x2, _packed(wq, wk, wv), _packed(bq, bk, bv))      # one [M, 3H] projection (2304 columns fill the chip; 3 x 768 do not)
    if qk_gate is not None:                                          # ViLBERT dynamic_attention gates on the Q | K columns (vilbert.py:211-212)
        nat.rowgroup_scale_f32(qkv, 3 * H, qk_gate.contiguous(), B, S, 2 * H)
    ctx = torch.empty(M, H, dtype=F32, device=dev)
    mask = _attn_mask(mask_add, B, S)
    nat.attention_f32_fwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, mask, ctx, H, B, heads, S, S, 1.0 / math.sqrt(hd), head_dim=hd,
                          causal_tail=int(causal_tail))
    y1 = _linear(ctx, wo, bo, resid=x2)
    a_out = torch.empty(M, H, dtype=F32, device=dev)
    nat.layernorm_f32_fwd(y1, _w(ln1_w), _w(ln1_b), a_out, M, H, eps1)
    hh = _linear(a_out, w1, b1, act=1)
    y2 = _linear(hh, w2, b2, resid=a_out)
    out = torch.empty(M, H, dtype=F32, device=dev)
    nat.layernorm_f32_fwd(y2, _w(ln2_w), _w(ln2_b), out, M, H, eps2)
    return out.view(B, S, H)


def masked_lm_head(x, weight, bias, labels, ignore_index):
    """Decoder tied to the word embeddings + CrossEntropyLoss(ignore_index) (visual_bert.py:267-277), forward only:
    (loss, logits [B, S, vocab])."""
    x2 = _rows(x)
    M = x2.shape[0]
    N = weight.shape[0]
    logits = _linear(x2, weight, bias)
    lab = labels.reshape(M).contiguous()
    lse = torch.empty(M, dtype=F32, device=x2.device)
    rowloss = torch.empty(M, dtype=F32, device=x2.device)
    loss = torch.empty(1, dtype=F32, device=x2.device)
    count = torch.empty(1, dtype=F32, device=x2.device)
    nat.vocab_cross_entropy_fwd(logits, lab, lse, rowloss, loss, count, M, N, ignore_index)
    return loss[0], logits.view(*x.shape[:-1], N)


def masked_region_head(x, weight, bias, target, row_label):
    """ViLBERT's masked-region classification head (vilbert.py:846-858, 1150-1157, `visual_target: 0`), forward: decoder GEMM + masked soft-target
    KL divergence.  Returns (loss, scores)."""
    x2 = _rows(x)
    M = x2.shape[0]
    N = weight.shape[0]
    logits = _linear(x2, weight, bias)
    lab = row_label.reshape(M).long().contiguous()
    tgt = target.reshape(M, N).float().contiguous()
    e = lambda *sh: torch.empty(*sh, dtype=F32, device=x2.device)
    lse, tsum, rowloss, loss, count = e(M), e(M), e(M), e(1), e(1)
    nat.soft_target_kl_fwd(logits, tgt, lab, lse, tsum, rowloss, loss, count, M, N)
    return loss[0], logits.view(*x.shape[:-1], N)


def mmbt_embeddings(feats, input_ids, start_tok, end_tok, text_type_ids, modal_type, word, pos, typ, ln_w, ln_b, proj_w, proj_b, eps):
    """ModalEmbeddings.forward (mmf/models/mmbt.py:84-129) + the text BertEmbeddings (hf_layers.py:108-135), modal block first
    (mmbt.py:225), in one fp32 buffer [start token | N projected features | end token | T text]; one LayerNorm pass (the two of
    the reference share their parameters).  Same layout as functional.MMBTEmbeddingsFn."""
    B, N, D = feats.shape
    T = input_ids.shape[1]
    H = word.shape[1]
    s0 = 1 if start_tok is not None else 0
    L = N + s0 + (1 if end_tok is not None else 0)
    S = L + T
    dev = word.device
    y = torch.empty(B * S, H, dtype=F32, device=dev)
    wd, pd, td = _w(word), _w(pos), _w(typ)
    from mmf_amd.functional import mmbt_modal_types
    mt_st, mt_en, coladd, rowtab, rowidx, _, _ = mmbt_modal_types(modal_type, B, L, N, s0, dev, pd, td)
    if start_tok is not None:
        nat.embed_text_f32_fwd(start_tok.reshape(B, 1).contiguous(), mt_st, wd, pd, td, y, B, 1, S, H, 0, 0)
    if end_tok is not None:
        nat.embed_text_f32_fwd(end_tok.reshape(B, 1).contiguous(), mt_en, wd, pd, td, y, B, 1, S, H, s0 + N, s0 + N)
    nat.embed_text_f32_fwd(input_ids.contiguous(), text_type_ids.contiguous(), wd, pd, td, y, B, T, S, H, L, 0)
    f2 = feats.reshape(B * N, D)
    f2 = (f2 if f2.dtype == F32 else f2.float()).contiguous()
    if D % 4:
        raise ValueError("fp32 path: modal feature width (%d) must be a multiple of 4" % D)
    nat.gemm_f32(f2, _w(proj_w), y, B * N, H, D, D, D, H, bias=_w(proj_b), coladd=coladd, rowtab=rowtab,
                 rowidx=rowidx, rowtab_ld=H, grp=(N, S - N, s0))
    out = torch.empty(B * S, H, dtype=F32, device=dev)
    nat.layernorm_f32_fwd(y, _w(ln_w), _w(ln_b), out, B * S, H, eps)
    return out.view(B, S, H)


def add_pos_type(x, seg, pos, typ):
    """total = tok + pos_emb(arange(L)) + token_type_embeddings(segment_ids) (mmf/models/transformers/backends/huggingface.py:147-155)."""
    B, L, H = x.shape
    y = torch.empty(B * L, H, dtype=F32, device=x.device)
    sg = seg.contiguous() if (seg is not None and typ is not None) else None
    nat.rows_add_embed_f32(_rows(x), sg, None if pos is None else _w(pos), _w(typ) if sg is not None else None, y, B, L, L, H)
    return y.view(B, L, H)


def concat_rows(*xs):
    """torch.cat(list_embeddings, dim=1) (huggingface.py:159) of fp32 [B, L_m, H] blocks: strided copies by the row-copy kernel
    (fp32 rows moved as pairs of 16-bit words)."""
    B, _, H = xs[0].shape
    lens = [int(x.shape[1]) for x in xs]
    S = sum(lens)
    out = torch.empty(B, S, H, dtype=F32, device=xs[0].device)
    ob = out.view(torch.bfloat16).view(B * S, 2 * H)
    off = 0
    for x, L in zip(xs, lens):
        nat.copy_rows(_rows(x).view(torch.bfloat16), L, ob[off:], S, B, L, 2 * H)
        off += L
    return out


# ---------------------------------------------------------------------------------------------
# ViLBERT (mmf/models/vilbert.py) and UNITER (mmf/models/uniter.py) on the fp32 kernels
# ---------------------------------------------------------------------------------------------
def _pad_k(x2, K, KP):
    out = torch.empty(x2.shape[0], KP, dtype=F32, device=x2.device)
    nat.pad_rows_f32(x2, K, out, KP, x2.shape[0])
    return out


def small_k_linear(x, weight, bias, resid=None):
    """nn.Linear over a handful of input features (vilbert.py:906: 5-d, uniter.py:81: 7-d box geometry): operand and weight rows
    zero-padded to a multiple of 4 columns (16-byte rows for the fp32 GEMM loader)."""
    K = x.shape[-1]
    KP = (K + 3) // 4 * 4
    x2 = x.reshape(-1, K)
    x2 = (x2 if x2.dtype == F32 else x2.float()).contiguous()
    w = _w(weight)
    N = w.shape[0]
    if KP != K:
        x2, w = _pad_k(x2, K, KP), _pad_k(w, K, KP)
    M = x2.shape[0]
    out = torch.empty(M, N, dtype=F32, device=x2.device)
    nat.gemm_f32(x2, w, out, M, N, KP, KP, KP, N, bias=_w(bias), resid=resid, ldr=0 if resid is None else resid.stride(0))
    return out.view(*x.shape[:-1], N)


def image_feature_embeddings(feats, loc, w_img, b_img, w_loc, b_loc, ln_w, ln_b, eps):
    """BertImageFeatureEmbeddings.forward (vilbert.py:904-913), eval mode: LayerNorm(Linear(features) + Linear(5-d location))."""
    B, R, D = feats.shape
    f2 = feats.reshape(B * R, D)
    f2 = (f2 if f2.dtype == F32 else f2.float()).contiguous()
    y0 = _linear(f2, w_img, b_img)
    y = small_k_linear(loc.reshape(B * R, loc.shape[-1]), w_loc, b_loc, resid=y0)
    out = torch.empty_like(y)
    nat.layernorm_f32_fwd(y, _w(ln_w), _w(ln_b), out, B * R, y.shape[1], eps)
    return out.view(B, R, -1)


def bi_attention(img, txt, q1, k1, v1, q2, k2, v2, img_mask_add, txt_mask_add, heads):
    """BertBiAttention.forward (vilbert.py:388-475), eval mode: (context_layer1 [B, T, bi] = text queries over image keys / values under
    the image mask, context_layer2 [B, R, bi] = image queries over text keys / values under the text mask).  q1..v2 are the six
    nn.Linear modules; each stream's Q | K | V is one packed fp32 GEMM."""
    B, R, _ = img.shape
    T = txt.shape[1]
    BH = q1.weight.shape[0]
    hd = BH // heads
    _check_head(hd, max(R, T))
    qkv1 = _linear(_rows(img), _packed(q1.weight, k1.weight, v1.weight), _packed(q1.bias, k1.bias, v1.bias))
    qkv2 = _linear(_rows(txt), _packed(q2.weight, k2.weight, v2.weight), _packed(q2.bias, k2.bias, v2.bias))
    dev = qkv1.device
    scale = 1.0 / math.sqrt(hd)
    ctx1 = torch.empty(B * T, BH, dtype=F32, device=dev)
    nat.attention_f32_fwd(qkv2, qkv1[:, BH:], qkv1[:, 2 * BH:], 3 * BH, 3 * BH, 3 * BH, img_mask_add.reshape(B, R).contiguous(), ctx1, BH,
                          B, heads, T, R, scale, head_dim=hd)
    ctx2 = torch.empty(B * R, BH, dtype=F32, device=dev)
    nat.attention_f32_fwd(qkv1, qkv2[:, BH:], qkv2[:, 2 * BH:], 3 * BH, 3 * BH, 3 * BH, txt_mask_add.reshape(B, T).contiguous(), ctx2, BH,
                          B, heads, R, T, scale, head_dim=hd)
    return ctx1.view(B, T, BH), ctx2.view(B, R, BH)


def dense_residual_ln(h, resid, weight, bias, ln_w, ln_b, eps):
    """BertSelfOutput / BertBiOutput halves: LayerNorm(dense(h) + resid) (hf_layers.py:245-252, vilbert.py:497-512), eval mode."""
    r2 = _rows(resid)
    y = _linear(_rows(h), weight, bias, resid=r2)
    out = torch.empty_like(y)
    nat.layernorm_f32_fwd(y, _w(ln_w), _w(ln_b), out, y.shape[0], y.shape[1], eps)
    return out.view(resid.shape)


def feed_forward(x, w1, b1, w2, b2, ln_w, ln_b, eps):
    """BertIntermediate + BertOutput (hf_layers.py:286-292): LayerNorm(dense2(gelu(dense1(x))) + x), eval mode."""
    x2 = _rows(x)
    hh = _linear(x2, w1, b1, act=1)
    y = _linear(hh, w2, b2, resid=x2)
    out = torch.empty_like(y)
    nat.layernorm_f32_fwd(y, _w(ln_w), _w(ln_b), out, y.shape[0], y.shape[1], eps)
    return out.view(x.shape)


def attention_block(x, wq, bq, wk, bk, wv, bv, wo, bo, ln_w, ln_b, mask_add, heads, eps, qk_gate=None):
    """BertAttentionJit.forward (hf_layers.py:233-252), eval mode, with ViLBERT's optional Q | K gates."""
    B, S, H = x.shape
    hd = H // heads
    _check_head(hd, S)
    x2 = _rows(x)
    qkv = _linear(x2, _packed(wq, wk, wv), _packed(bq, bk, bv))
    if qk_gate is not None:
        nat.rowgroup_scale_f32(qkv, 3 * H, qk_gate.contiguous(), B, S, 2 * H)
    ctx = torch.empty(B * S, H, dtype=F32, device=x2.device)
    nat.attention_f32_fwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, None if mask_add is None else mask_add.reshape(B, S).contiguous(),
                          ctx, H, B, heads, S, S, 1.0 / math.sqrt(hd), head_dim=hd)
    return dense_residual_ln(ctx.view(B, S, H), x, wo, bo, ln_w, ln_b, eps)


def masked_mean(x, mask):
    """(x * mask.unsqueeze(-1)).sum(1) / mask.sum(1, keepdim=True) (vilbert.py:204-205): [B, T, H], [B, T] -> [B, H]."""
    B, T, H = x.shape
    pool = torch.empty(B, H, dtype=F32, device=x.device)
    nat.masked_mean_f32(_rows(x), mask.reshape(B, T).float().contiguous(), pool, B, T, H)
    return pool


def dynamic_gate(zq, zk):
    """cat([1 + sigmoid(zq), 1 + sigmoid(zk)], dim=1) (vilbert.py:206-209) — the gate kernel of the throughput path is fp32 already."""
    B, Cn = zq.shape
    gate = torch.empty(B, 2 * Cn, dtype=F32, device=zq.device)
    nat.gate_sigmoid_fwd(zq.contiguous(), gate, 0, B, Cn)
    nat.gate_sigmoid_fwd(zk.contiguous(), gate, Cn, B, Cn)
    return gate


def _eltwise(op, a, b=None):
    a2 = _rows(a)
    out = torch.empty_like(a2)
    nat.eltwise_f32(op, a2, None if b is None else _rows(b), out)
    return out.view(a.shape)


def expand_batch(x, reps, mode):
    """ViLBERT's `in_batch_pairs` / `fast_mode` batch expansion (vilbert.py:678-725), forward only: [Bs, L, H] -> [reps * Bs, L, H]."""
    Bs, L, H = x.shape
    x2 = _rows(x)
    out = torch.empty(reps * Bs * L, H, dtype=F32, device=x2.device)
    nat.expand_batch(x2, out, Bs, reps, L * H, mode)
    return out.view(reps * Bs, L, H)


def eltwise_mul(a, b):
    return _eltwise(0, a, b)


def relu(a):
    return _eltwise(1, a)


def add(a, b):
    return _eltwise(3, a, b)


def feature_table_add(feats, idx, table):
    """img_feat + mask_embedding(img_masks) (uniter.py:74-78) on fp32 rows; without masks the features pass through."""
    B, R, D = feats.shape
    f2 = feats.reshape(B * R, D)
    f2 = (f2 if f2.dtype == F32 else f2.float()).contiguous()
    if idx is None:
        return f2.view(B, R, D)
    y = torch.empty(B * R, D, dtype=F32, device=f2.device)
    nat.rows_add_embed_f32(f2, idx.reshape(B, R).long().contiguous(), None, _w(table), y, B, R, R, D)
    return y.view(B, R, D)


# ---------------------------------------------------------------------------------------------
# M4C (mmf/models/m4c.py) on the fp32 kernels
# ---------------------------------------------------------------------------------------------
def l2norm_rows(x):
    """F.normalize(x, dim=-1) (m4c.py:195)."""
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    x2 = (x2 if x2.dtype == F32 else x2.float()).contiguous()
    y = torch.empty_like(x2)
    nat.l2norm_rows_f32(x2, D, y, D, x2.shape[0], D)
    return y.view(x.shape)


def ocr_feature_concat(fasttext, phoc, fc7, order_dim):
    """cat([normalize(fasttext), normalize(phoc), normalize(fc7), zeros(order vectors)], -1) (m4c.py:211-237) written once as fp32 rows
    padded to a multiple of 4 columns (16-byte rows for the GEMM loader)."""
    B, N, _ = fasttext.shape
    rows = B * N
    d0, d1, d2 = fasttext.shape[-1], phoc.shape[-1], fc7.shape[-1]
    K = d0 + d1 + d2 + int(order_dim)
    KP = (K + 3) // 4 * 4
    out = torch.zeros(rows, KP, dtype=F32, device=fc7.device)
    off = 0
    for f, d in ((fasttext, d0), (phoc, d1), (fc7, d2)):
        f2 = f.reshape(rows, d)
        f2 = (f2 if f2.dtype == F32 else f2.float()).contiguous()
        nat.l2norm_rows_f32(f2, d, out[:, off:], KP, rows, d)
        off += d
    return out.view(B, N, KP), K


def padded_linear(x, weight, bias):
    """nn.Linear on rows already zero-padded to a multiple of 4 columns (the 3002-wide OCR feature, m4c.py:243)."""
    K = weight.shape[1]
    KP = x.shape[-1]
    x2 = _rows(x)
    w = _w(weight)
    if KP != K:
        w = _pad_k(w, K, KP)
    M = x2.shape[0]
    out = torch.empty(M, w.shape[0], dtype=F32, device=x2.device)
    nat.gemm_f32(x2, w, out, M, w.shape[0], KP, KP, KP, w.shape[0], bias=_w(bias))
    return out.view(*x.shape[:-1], w.shape[0])


def prev_pred_gather(ans, ocr, prev_inds):
    """_batch_gather(cat([ans_emb.expand(B), ocr_emb], 1), prev_inds) (m4c.py:526-528) as one two-source row gather."""
    V, H = ans.shape
    B, N, _ = ocr.shape
    T = prev_inds.shape[1]
    batch = torch.arange(B, device=prev_inds.device, dtype=torch.int64).unsqueeze(1) * N
    flat = torch.where(prev_inds < V, prev_inds, prev_inds + batch).contiguous()
    out = torch.empty(B * T, H, dtype=F32, device=ocr.device)
    nat.gather_rows2_f32(_rows(ans), _rows(ocr), flat, out, B * T, H)
    return out.view(B, T, H)


def split_rows(x, lens):
    """torch.split along dim 1 into contiguous fp32 blocks (m4c.py:446-449)."""
    B, S, H = x.shape
    xb = _rows(x).view(torch.bfloat16)
    outs, off = [], 0
    for L in lens:
        d = torch.empty(B * L, H, dtype=F32, device=x.device)
        nat.copy_rows(xb[off:], S, d.view(torch.bfloat16), L, B, L, 2 * H)
        outs.append(d.view(B, L, H))
        off += L
    return tuple(outs)


def m4c_scores(dec, ocr, cls_w, cls_b, q_w, q_b, k_w, k_b, ocr_mask_add):
    """M4C._forward_output (m4c.py:275-283): classifier scores and OCR pointer scores written into one fp32 [B, T, V + N] buffer."""
    B, T, H = dec.shape
    N = ocr.shape[1]
    V, HQ = cls_w.shape[0], q_w.shape[0]
    d2, o2 = _rows(dec), _rows(ocr)
    out = torch.empty(B * T, V + N, dtype=F32, device=d2.device)
    nat.gemm_f32(d2, _w(cls_w), out, B * T, V, H, H, H, V + N, bias=_w(cls_b))
    q = _linear(d2, q_w, q_b)
    k = _linear(o2, k_w, k_b)
    nat.ptr_scores_f32(q, k, ocr_mask_add.reshape(B, N).float().contiguous(), out[:, V:], V + N, B, T, N, HQ, 1.0 / math.sqrt(HQ))
    return out.view(B, T, V + N)
