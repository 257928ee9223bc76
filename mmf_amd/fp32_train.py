"""fp32 TRAINING on the fp32 kernels: the reference's default arithmetic for a training step.

The reference computes forward AND backward in fp32 unless `training.fp16` turns autocast on (mmf/trainers/core/training_loop.py:199-211).
The throughput path is bf16 (north_star's 5e-2 bound); `mmf_amd.fp32_inference()` is the fp32 forward.  Inside

    with mmf_amd.fp32_training():
        out = model(sample_list)            # train or eval mode; dropout works
    out["losses"][...].backward()           # the backward kernels are launched by the autograd nodes built above
    optimizer.step()                        # the fused AdamW takes the fp32 gradients as they are

every `torch.ops.mmf_amd.*` operator of the VisualBERT training step builds an autograd node whose forward and backward run on
mmf_amd/csrc/fp32_path.hip (`mmf_gemm_f32` in its forward / dgrad / weight-gradient layouts, `mmf_attention_f32_fwd` / `_bwd`) and
mmf_amd/csrc/fp32_train.hip (LayerNorm backward, column sums, dropout, row scatters): fp32 activations, fp32 master parameters read
directly, fp32 gradients.  Parity: every parameter gradient of the reference's fixture and of the full VisualBERT-base VQA2 configuration
against the CPU oracle within north_star's fp32 bound (tests/test_fp32_train_gpu.py).

Built for the operators the classification steps of VisualBERT, ViLBERT, MMBT, the MMF Transformer and UNITER use (their embedding stages, encoder
layers, ViLBERT's co-attention and output blocks, poolers, nlvr2 pairing, prediction-head transform, classifiers, logit_bce; cross_entropy is
fp32 already); operators outside that set raise NotImplementedError inside the context instead of silently dropping to bf16.

Reference operations, as in mmf_amd/functional.py: BertVisioLinguisticEmbeddings.forward (mmf/modules/embeddings.py:423-459), BertLayerJit
.forward (mmf/modules/hf_layers.py:255-292), BertPooler / BertPredictionHeadTransform / classifier Linear (mmf/models/visual_bert.py:146,
327-330, 389-401), LogitBinaryCrossEntropy (mmf/modules/losses.py:246-251)."""
import contextlib
import math

import torch

from mmf_amd import _native as nat
from mmf_amd import fp32_path as P

F32 = torch.float32
_depth = 0


def active():
    return _depth > 0


@contextlib.contextmanager
def fp32_training():
    """Run every mmf_amd operator inside the block on the fp32 kernels WITH autograd (forward here, backward when `.backward()` runs)."""
    global _depth
    from mmf_amd import _ops_native
    _depth += 1
    _ops_native.push_mode(0)        # the native operators forward to their Python twins, which route here
    try:
        yield
    finally:
        _depth -= 1
        _ops_native.pop_mode(0)


def make_drop(p, training):
    """Dropout configuration of one site: the key stream of the throughput path (functional.dropout_keys: one base key per step from torch's
    generator, a host-side counter per site — reproducible under torch.manual_seed, no device read-back per site)."""
    from mmf_amd import functional as Fn
    return Fn.make_drop(p, training)


def _rows(x):
    return P._rows(x)


def _w(p):
    return P._w(p)


def _empty(*shape, like):
    return torch.empty(*shape, dtype=F32, device=like.device)


def _gemm(x2, w, out_cols, bias=None, **kw):
    """x2 [M, K] @ w[N, K]^T (+ epilogue) -> [M, N]"""
    M, K = x2.shape
    out = _empty(M, out_cols, like=x2)
    nat.gemm_f32(x2, w, out, M, out_cols, K, x2.stride(0), w.stride(0), out_cols, bias=bias, **kw)
    return out


def _dgrad(dz, w, resid=None, act=0, aux=None):
    """dX = dZ W (+ resid): dZ [M, N], W [N, K] as stored -> [M, K]; optional saved-derivative multiply in the epilogue."""
    M, N = dz.shape
    K = w.shape[1]
    out = _empty(M, K, like=dz)
    nat.gemm_f32(dz, w, out, M, K, N, dz.stride(0), w.stride(0), K, b_kmajor=True, resid=resid, ldr=0 if resid is None else resid.stride(0),
                 act=act, aux=aux)
    return out


def _wgrad(dz, x2):
    """dW = dZ^T X: dZ [M, N], X [M, K] -> [N, K]; split-K over the M rows."""
    M, N = dz.shape
    K = x2.shape[1]
    out = _empty(N, K, like=dz)
    nat.gemm_f32(dz, x2, out, N, K, M, dz.stride(0), x2.stride(0), K, a_kmajor=True, b_kmajor=True, split_k=True)
    return out


def _colsum(dz):
    M, N = dz.shape
    out = _empty(N, like=dz)
    nat.colsum_f32(dz, dz.stride(0), M, N, out)
    return out



def _ln_fwd(y, gamma, beta, eps):
    M, H = y.shape
    out = _empty(M, H, like=y)
    mean = _empty(M, like=y); rstd = _empty(M, like=y)
    nat.layernorm_f32_fwd_stats(y, _w(gamma), _w(beta), out, mean, rstd, M, H, eps)
    return out, mean, rstd


def _ln_bwd(g2, y, mean, rstd, gamma):
    M, H = y.shape
    dx = _empty(M, H, like=y); dg = _empty(H, like=y); db = _empty(H, like=y)
    nat.layernorm_f32_bwd(g2, y, mean, rstd, _w(gamma), dx, dg, db, M, H)
    return dx, dg, db


def _drop_bwd(g2, drop):
    if not drop[1]:
        return g2
    out = torch.empty_like(g2)
    nat.dropout_f32(g2, out, drop)
    return out


def _grad2(g, cols):
    g2 = g.reshape(-1, cols)
    g2 = g2 if g2.dtype == F32 else g2.float()
    return g2 if g2.is_contiguous() else g2.contiguous()


def _pad4(dz):
    """A gradient whose width is not a multiple of 4 (the 3129 answer logits) as a zero-padded buffer with 16-byte rows; the returned
    view keeps the true width, its stride is the padded one."""
    M, N = dz.shape
    if N % 4 == 0:
        return dz
    NP = (N + 3) // 4 * 4
    out = _empty(M, NP, like=dz)
    nat.pad_rows_f32(dz, N, out, NP, M)
    return out[:, :N]


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b): act 0 none, 1 exact-erf GELU (its derivative saved by the epilogue), 3 tanh."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x2 = _rows(x)
        w = _w(weight)
        N = w.shape[0]
        if x2.shape[1] % 4:
            raise ValueError("fp32 path: the contraction length (%d) must be a multiple of 4" % x2.shape[1])
        U = _empty(x2.shape[0], N, like=x2) if act == 1 else None
        y = _gemm(x2, w, N, bias=None if bias is None else _w(bias), act=act, U=U)
        ctx.save_for_backward(x2, w, U if act == 1 else (y if act == 3 else None))
        ctx.meta = (x.shape, act, bias is not None)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, w, aux = ctx.saved_tensors
        xshape, act, has_bias = ctx.meta
        dz = _grad2(g, w.shape[0])
        if act == 1 or act == 3:
            t = torch.empty_like(dz)
            nat.eltwise_f32(0 if act == 1 else 4, dz, aux, t)       # dy * gelu'(pre)  |  dy * (1 - y^2)
            dz = t
        dz = _pad4(dz)
        dx = _dgrad(dz, w).view(xshape) if ctx.needs_input_grad[0] else None
        return dx, _wgrad(dz, x2), (_colsum(dz) if has_bias else None), None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x2 = _rows(x)
        out, mean, rstd = _ln_fwd(x2, gamma, beta, eps)
        ctx.save_for_backward(x2, mean, rstd, gamma.detach())
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        x2, mean, rstd, gamma = ctx.saved_tensors
        dx, dg, db = _ln_bwd(_grad2(g, x2.shape[1]), x2, mean, rstd, gamma)
        return dx.view(g.shape), dg, db, None


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, drop):
        x2 = _rows(x)
        y = torch.empty_like(x2)
        nat.dropout_f32(x2, y, drop)
        ctx.drop = drop
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        return _drop_bwd(_grad2(g, g.shape[-1]), ctx.drop).view(g.shape), None


class GatherRowsFn(torch.autograd.Function):
    """out[b] = x[b, index[b]] (the `vqa` pooling, visual_bert.py:389-398)."""

    @staticmethod
    def forward(ctx, x, index):
        B, S, H = x.shape
        out = _empty(B, H, like=x)
        ix = index.contiguous().long()
        nat.gather_rows_f32(_rows(x), ix, out, B, S, H)
        ctx.save_for_backward(ix)
        ctx.meta = (B, S, H)
        return out

    @staticmethod
    def backward(ctx, g):
        (ix,) = ctx.saved_tensors
        B, S, H = ctx.meta
        dx = torch.zeros(B * S, H, dtype=F32, device=g.device)
        nat.scatter_add_rows_f32(_grad2(g, H), H, B, H, ix, dx, H, dst_stride=S)
        return dx.view(B, S, H), None


class AttentionBlockFn(torch.autograd.Function):
    """BertAttentionJit.forward (hf_layers.py:233-252): packed Q|K|V projection, fused attention, output projection + bias + dropout +
    residual in one GEMM epilogue, LayerNorm.  Backward: LayerNorm, dropout, the output projection's dgrad / wgrad, the attention
    backward into one [M, 3H] buffer, and the Q|K|V dgrad with the residual gradient added in ITS epilogue."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, mask_add, heads, eps, drop_attn, drop_hid, tail, qk_gate=None):
        B, S, H = x.shape
        hd = H // heads
        P._check_head(hd, S)
        x2 = _rows(x)
        M = B * S
        wqkv, bqkv = P._packed(wq, wk, wv), P._packed(bq, bk, bv)
        qkv = _gemm(x2, wqkv, 3 * H, bias=bqkv)
        gate = None
        if qk_gate is not None:      # ViLBERT dynamic_attention: per-sample column gates on the Q | K columns (vilbert.py:211-212)
            gate = qk_gate.detach().float().contiguous()
            nat.rowgroup_scale_f32(qkv, 3 * H, gate, B, S, 2 * H)
        ctxt = _empty(M, H, like=x2)
        lse = _empty(B, heads, S, like=x2)
        mask = P._attn_mask(mask_add, B, S)
        nat.attention_f32_fwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, mask, ctxt, H, B, heads, S, S, 1.0 / math.sqrt(hd), head_dim=hd,
                              causal_tail=int(tail), lse=lse, drop=drop_attn)
        wo_ = _w(wo)
        y1 = _gemm(ctxt, wo_, H, bias=_w(bo), drop=drop_hid, resid=x2, ldr=H)
        out, mean, rstd = _ln_fwd(y1, gamma, beta, eps)
        ctx.save_for_backward(x2, qkv, ctxt, lse, y1, mean, rstd, wqkv.clone(), wo_, gamma.detach(), mask, gate)
        ctx.meta = (B, S, H, heads, drop_attn, drop_hid, int(tail))
        return out.view(B, S, H)

    @staticmethod
    def backward(ctx, g):
        x2, qkv, ctxt, lse, y1, mean, rstd, wqkv, wo, gamma, mask, gate = ctx.saved_tensors
        B, S, H, heads, drop_attn, drop_hid, tail = ctx.meta
        hd = H // heads
        M = B * S
        dy1, dgamma, dbeta = _ln_bwd(_grad2(g, H), y1, mean, rstd, gamma)
        dz = _drop_bwd(dy1, drop_hid)
        dctx = _dgrad(dz, wo)
        dwo, dbo = _wgrad(dz, ctxt), _colsum(dz)
        dqkv = _empty(M, 3 * H, like=x2)
        delta = _empty(B, heads, S, like=x2)
        nat.attention_f32_bwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, mask, ctxt, H, lse, B, heads, S, S, 1.0 / math.sqrt(hd), dctx,
                              dqkv, dqkv[:, H:], dqkv[:, 2 * H:], delta, head_dim=hd, causal_tail=tail, drop=drop_attn)
        dgate = None
        if gate is not None:         # dqkv becomes the gradient of the un-gated projection; dgate[b][c] = sum_rows dqk * (un-gated qk)
            dgate = torch.empty_like(gate)
            nat.rowgroup_scale_f32_bwd(dqkv, qkv, 3 * H, gate, dgate, B, S, 2 * H)
        dx = _dgrad(dqkv, wqkv, resid=dy1)
        dw, db = _wgrad(dqkv, x2), _colsum(dqkv)
        return (dx.view(B, S, H), dw[:H], db[:H], dw[H:2 * H], db[H:2 * H], dw[2 * H:], db[2 * H:], dwo, dbo, dgamma, dbeta,
                None, None, None, None, None, None, dgate)


class FeedForwardFn(torch.autograd.Function):
    """BertIntermediate + BertOutput (hf_layers.py:286-292): GELU in the up-projection epilogue (its derivative saved), down-projection +
    bias + dropout + residual, LayerNorm.  Backward: the down-projection's dgrad multiplies by the saved derivative in its epilogue,
    the up-projection's dgrad adds the residual gradient in its own."""

    @staticmethod
    def forward(ctx, a, w1, b1, w2, b2, gamma, beta, eps, drop):
        B, S, H = a.shape
        a2 = _rows(a)
        M = B * S
        w1_, w2_ = _w(w1), _w(w2)
        I = w1_.shape[0]
        U = _empty(M, I, like=a2)
        hh = _gemm(a2, w1_, I, bias=_w(b1), act=1, U=U)
        y2 = _gemm(hh, w2_, H, bias=_w(b2), drop=drop, resid=a2, ldr=H)
        out, mean, rstd = _ln_fwd(y2, gamma, beta, eps)
        ctx.save_for_backward(a2, hh, U, y2, mean, rstd, w1_, w2_, gamma.detach())
        ctx.meta = (B, S, H, drop)
        return out.view(B, S, H)

    @staticmethod
    def backward(ctx, g):
        a2, hh, U, y2, mean, rstd, w1, w2, gamma = ctx.saved_tensors
        B, S, H, drop = ctx.meta
        dy2, dgamma, dbeta = _ln_bwd(_grad2(g, H), y2, mean, rstd, gamma)
        dz2 = _drop_bwd(dy2, drop)
        dpre = _dgrad(dz2, w2, act=2, aux=U)                    # d(hh) * gelu'(pre)
        dw2, db2 = _wgrad(dz2, hh), _colsum(dz2)
        da = _dgrad(dpre, w1, resid=dy2)
        dw1, db1 = _wgrad(dpre, a2), _colsum(dpre)
        return da.view(B, S, H), dw1, db1, dw2, db2, dgamma, dbeta, None, None


class VisioLinguisticEmbeddingsFn(torch.autograd.Function):
    """BertVisioLinguisticEmbeddings.forward (embeddings.py:423-459): text rows = word + position + type, visual rows = projection(features)
    + visual type + visual position 0 written by the projection GEMM's epilogue into rows T.. of the joint sequence; LayerNorm; dropout.
    Backward: dropout, LayerNorm, row scatters into the three text tables (the padding row of the word table dropped) and the visual
    type table, the projection's weight gradient over the gathered visual rows."""

    @staticmethod
    def forward(ctx, input_ids, token_type_ids, feats, vtype, word, pos, typ, ln_w, ln_b, typ_vis, pos_vis, proj_w, proj_b, eps, drop, pad_idx,
                align=None):
        B, T = input_ids.shape
        H = word.shape[1]
        R = 0 if feats is None else feats.shape[1]
        S = T + R
        dev = word.device
        ids, seg = input_ids.contiguous(), token_type_ids.contiguous()
        y = torch.empty(B * S, H, dtype=F32, device=dev)
        nat.embed_text_f32_fwd(ids, seg, _w(word), _w(pos), _w(typ), y, B, T, S, H)
        f2 = vt = None
        if R:
            D = feats.shape[2]
            if D % 4:
                raise ValueError("fp32 path: visual feature width (%d) must be a multiple of 4" % D)
            f2 = feats.reshape(B * R, D)
            f2 = (f2 if f2.dtype == F32 else f2.float()).contiguous()
            vt = vtype.reshape(B * R).contiguous()
            if align is not None:    # image_text_alignment (embeddings.py:373-397): per region, the mean TEXT position row of its aligned words + visual type
                al = align.reshape(B * R, -1).long().contiguous()
                addend = torch.empty(B * R, H, dtype=F32, device=dev)
                nat.align_pos_fwd(al, _w(pos), _w(typ_vis), vt, addend, B * R, al.shape[1], H)
                nat.gemm_f32(f2, _w(proj_w), y, B * R, H, D, D, D, H, bias=_w(proj_b), coladd=_w(pos_vis)[0], rowtab=addend,
                             rowidx=torch.arange(B * R, dtype=torch.int64, device=dev), rowtab_ld=H, grp=(R, T, T))
            else:
                al = None
                nat.gemm_f32(f2, _w(proj_w), y, B * R, H, D, D, D, H, bias=_w(proj_b), coladd=_w(pos_vis)[0], rowtab=_w(typ_vis), rowidx=vt,
                             rowtab_ld=H, grp=(R, T, T))
        else:
            al = None
        out, mean, rstd = _ln_fwd(y, ln_w, ln_b, eps)
        if drop[1]:
            o2 = torch.empty_like(out)
            nat.dropout_f32(out, o2, drop)
            out = o2
        ctx.save_for_backward(ids, seg, f2, vt, y, mean, rstd, ln_w.detach(), al)
        ctx.meta = (B, T, R, H, drop, pad_idx, word.shape[0], pos.shape[0], typ.shape[0], 0 if typ_vis is None else typ_vis.shape[0],
                    0 if pos_vis is None else pos_vis.shape[0])
        return out.view(B, S, H)

    @staticmethod
    def backward(ctx, g):
        ids, seg, f2, vt, y, mean, rstd, ln_w, al = ctx.saved_tensors
        B, T, R, H, drop, pad_idx, V, NP, NT, NTV, NPV = ctx.meta
        S = T + R
        dev = y.device
        dy, dln_w, dln_b = _ln_bwd(_drop_bwd(_grad2(g, H), drop), y, mean, rstd, ln_w)
        dword = torch.zeros(V, H, dtype=F32, device=dev)
        nat.scatter_add_rows_f32(dy, H, B * T, H, ids.reshape(-1), dword, H, grp=(T, S, 0), skip=-1 if pad_idx is None else int(pad_idx))
        dpos = torch.zeros(NP, H, dtype=F32, device=dev)
        nat.scatter_add_rows_f32(dy, H, B * T, H, torch.arange(T, device=dev).repeat(B), dpos, H, grp=(T, S, 0))
        dtyp = torch.zeros(NT, H, dtype=F32, device=dev)
        nat.scatter_add_rows_f32(dy, H, B * T, H, seg.reshape(-1), dtyp, H, grp=(T, S, 0))
        dtv = dpv = dpw = dpb = None
        if R:
            dvis = torch.empty(B * R, H, dtype=F32, device=dev)      # the visual rows of dy, gathered (fp32 rows moved as pairs of 16-bit words)
            nat.copy_rows(dy.view(torch.bfloat16)[T:], S, dvis.view(torch.bfloat16), R, B, R, 2 * H)
            dpw = _wgrad(dvis, f2)
            dpb = _colsum(dvis)
            dtv = torch.zeros(NTV, H, dtype=F32, device=dev)
            nat.scatter_add_rows_f32(dvis, H, B * R, H, vt, dtv, H)
            dpv = torch.zeros(NPV, H, dtype=F32, device=dev)
            dpv[0].copy_(dpb)                                        # every visual row takes position row 0 (embeddings.py:411-418)
            if al is not None:                                       # the aligned words' TEXT position rows collect the regions' gradients / count
                nat.align_pos_f32_bwd(dvis, H, B, R, R, al, dpos, al.shape[1], H)
        return None, None, None, None, dword, dpos, dtyp, dln_w, dln_b, dtv, dpv, dpw, dpb, None, None, None, None


class PairHalvesFn(torch.autograd.Function):
    """nlvr2 pairing [2B, H] -> [B, 2H] = cat(x[:B], x[B:], dim=1) (visual_bert.py:369-374); backward = the two row copies reversed."""

    @staticmethod
    def forward(ctx, x):
        return P.pair_halves(x)

    @staticmethod
    def backward(ctx, g):
        g2 = _grad2(g, g.shape[-1])
        B, H2 = g2.shape
        H = H2 // 2
        dx = _empty(2 * B, H, like=g2)
        gb = g2.view(torch.bfloat16).view(2 * B, 2 * H)               # fp32 rows moved as pairs of 16-bit words: row 2b | 2b + 1 = the halves of sample b
        db = dx.view(torch.bfloat16)
        nat.copy_rows(gb, 2, db, 1, B, 1, 2 * H)
        nat.copy_rows(gb[1:], 2, db[B:], 1, B, 1, 2 * H)
        return dx


# ---- ViLBERT (mmf/models/vilbert.py) -----------------------------------------------------------------------------------------------------
class DenseResidualLNFn(torch.autograd.Function):
    """LayerNorm(dropout(dense(h)) + resid): BertSelfOutput / the two halves of BertBiOutput (hf_layers.py:245-252, vilbert.py:497-512); `h` and
    `resid` may have different widths."""

    @staticmethod
    def forward(ctx, h, resid, weight, bias, gamma, beta, eps, drop):
        h2, r2 = _rows(h), _rows(resid)
        w = _w(weight)
        N = w.shape[0]
        y = _gemm(h2, w, N, bias=_w(bias), drop=drop, resid=r2, ldr=r2.stride(0))
        out, mean, rstd = _ln_fwd(y, gamma, beta, eps)
        ctx.save_for_backward(h2, w, y, mean, rstd, gamma.detach())
        ctx.meta = (h.shape, resid.shape, drop)
        return out.view(resid.shape)

    @staticmethod
    def backward(ctx, g):
        h2, w, y, mean, rstd, gamma = ctx.saved_tensors
        hshape, rshape, drop = ctx.meta
        dy, dgamma, dbeta = _ln_bwd(_grad2(g, y.shape[1]), y, mean, rstd, gamma)
        dz = _drop_bwd(dy, drop)
        return _dgrad(dz, w).view(hshape), dy.view(rshape), _wgrad(dz, h2), _colsum(dz), dgamma, dbeta, None, None


class BiAttentionFn(torch.autograd.Function):
    """BertBiAttention.forward (vilbert.py:388-475): each stream's Q | K | V as one packed fp32 GEMM; context_layer1 = text queries over image
    keys / values (image mask, dropout1), context_layer2 = image queries over text keys / values (text mask, dropout2).  Backward: the two
    attention backwards write disjoint column blocks of the two packed gradient buffers, then each stream's dgrad / wgrad."""

    @staticmethod
    def forward(ctx, img, txt, q1w, q1b, k1w, k1b, v1w, v1b, q2w, q2b, k2w, k2b, v2w, v2b, img_mask_add, txt_mask_add, heads, drop1, drop2):
        B, R, _ = img.shape
        T = txt.shape[1]
        BH = q1w.shape[0]
        hd = BH // heads
        P._check_head(hd, max(R, T))
        i2, t2 = _rows(img), _rows(txt)
        w1, b1 = P._packed(q1w, k1w, v1w).clone(), P._packed(q1b, k1b, v1b)
        w2, b2 = P._packed(q2w, k2w, v2w).clone(), P._packed(q2b, k2b, v2b)
        qkv1 = _gemm(i2, w1, 3 * BH, bias=b1)
        qkv2 = _gemm(t2, w2, 3 * BH, bias=b2)
        scale = 1.0 / math.sqrt(hd)
        m1 = img_mask_add.reshape(B, R).float().contiguous()
        m2 = txt_mask_add.reshape(B, T).float().contiguous()
        ctx1 = _empty(B * T, BH, like=i2); lse1 = _empty(B, heads, T, like=i2)
        nat.attention_f32_fwd(qkv2, qkv1[:, BH:], qkv1[:, 2 * BH:], 3 * BH, 3 * BH, 3 * BH, m1, ctx1, BH, B, heads, T, R, scale, head_dim=hd, lse=lse1,
                              drop=drop1)
        ctx2 = _empty(B * R, BH, like=i2); lse2 = _empty(B, heads, R, like=i2)
        nat.attention_f32_fwd(qkv1, qkv2[:, BH:], qkv2[:, 2 * BH:], 3 * BH, 3 * BH, 3 * BH, m2, ctx2, BH, B, heads, R, T, scale, head_dim=hd, lse=lse2,
                              drop=drop2)
        ctx.save_for_backward(i2, t2, qkv1, qkv2, ctx1, ctx2, lse1, lse2, w1, w2, m1, m2)
        ctx.meta = (B, R, T, BH, heads, drop1, drop2, img.shape, txt.shape)
        return ctx1.view(B, T, BH), ctx2.view(B, R, BH)

    @staticmethod
    def backward(ctx, g1, g2):
        i2, t2, qkv1, qkv2, ctx1, ctx2, lse1, lse2, w1, w2, m1, m2 = ctx.saved_tensors
        B, R, T, BH, heads, drop1, drop2, ishape, tshape = ctx.meta
        hd = BH // heads
        scale = 1.0 / math.sqrt(hd)
        dqkv1 = _empty(B * R, 3 * BH, like=i2); dqkv2 = _empty(B * T, 3 * BH, like=i2)
        delta1 = _empty(B, heads, T, like=i2); delta2 = _empty(B, heads, R, like=i2)
        nat.attention_f32_bwd(qkv2, qkv1[:, BH:], qkv1[:, 2 * BH:], 3 * BH, 3 * BH, 3 * BH, m1, ctx1, BH, lse1, B, heads, T, R, scale, _grad2(g1, BH),
                              dqkv2, dqkv1[:, BH:], dqkv1[:, 2 * BH:], delta1, head_dim=hd, drop=drop1)
        nat.attention_f32_bwd(qkv1, qkv2[:, BH:], qkv2[:, 2 * BH:], 3 * BH, 3 * BH, 3 * BH, m2, ctx2, BH, lse2, B, heads, R, T, scale, _grad2(g2, BH),
                              dqkv1, dqkv2[:, BH:], dqkv2[:, 2 * BH:], delta2, head_dim=hd, drop=drop2)
        dimg, dw1, db1 = _dgrad(dqkv1, w1), _wgrad(dqkv1, i2), _colsum(dqkv1)
        dtxt, dw2, db2 = _dgrad(dqkv2, w2), _wgrad(dqkv2, t2), _colsum(dqkv2)
        return (dimg.view(ishape), dtxt.view(tshape),
                dw1[:BH], db1[:BH], dw1[BH:2 * BH], db1[BH:2 * BH], dw1[2 * BH:], db1[2 * BH:],
                dw2[:BH], db2[:BH], dw2[BH:2 * BH], db2[BH:2 * BH], dw2[2 * BH:], db2[2 * BH:],
                None, None, None, None, None)


class ImageFeatureEmbeddingsFn(torch.autograd.Function):
    """BertImageFeatureEmbeddings.forward (vilbert.py:904-913): LayerNorm(Linear(features) + Linear(5-d location)), dropout.  The inputs get no
    gradient (pre-extracted features)."""

    @staticmethod
    def forward(ctx, feats, loc, w_img, b_img, w_loc, b_loc, ln_w, ln_b, eps, drop):
        B, R, D = feats.shape
        f2 = feats.reshape(B * R, D)
        f2 = (f2 if f2.dtype == F32 else f2.float()).contiguous()
        KL = loc.shape[-1]
        KP = (KL + 3) // 4 * 4
        l2 = loc.reshape(B * R, KL)
        l2 = (l2 if l2.dtype == F32 else l2.float()).contiguous()
        lp = P._pad_k(l2, KL, KP) if KP != KL else l2
        wl = P._pad_k(_w(w_loc), KL, KP) if KP != KL else _w(w_loc)
        wi = _w(w_img)
        VH = wi.shape[0]
        y0 = _gemm(f2, wi, VH, bias=_w(b_img))
        y = _gemm(lp, wl, VH, bias=_w(b_loc), resid=y0, ldr=VH)
        out, mean, rstd = _ln_fwd(y, ln_w, ln_b, eps)
        if drop[1]:
            o2 = torch.empty_like(out)
            nat.dropout_f32(out, o2, drop)
            out = o2
        ctx.save_for_backward(f2, lp, y, mean, rstd, ln_w.detach())
        ctx.meta = (B, R, VH, KL, drop)
        return out.view(B, R, VH)

    @staticmethod
    def backward(ctx, g):
        f2, lp, y, mean, rstd, ln_w = ctx.saved_tensors
        B, R, VH, KL, drop = ctx.meta
        dy, dgamma, dbeta = _ln_bwd(_drop_bwd(_grad2(g, VH), drop), y, mean, rstd, ln_w)
        db = _colsum(dy)
        dwl = _wgrad(dy, lp)
        return None, None, _wgrad(dy, f2), db, dwl[:, :KL].contiguous(), db.clone(), dgamma, dbeta, None, None


class EltwiseFn(torch.autograd.Function):
    """op 0: a * b (vilbert.py:1318), op 1: relu(a) (the poolers, vilbert.py:803,818), op 3: a + b (vilbert.py:1320)."""

    @staticmethod
    def forward(ctx, op, a, b):
        a2 = _rows(a)
        b2 = None if b is None else _rows(b)
        y = torch.empty_like(a2)
        nat.eltwise_f32(op, a2, b2, y)
        ctx.op = op
        ctx.save_for_backward(a2, b2, y)
        return y.view(a.shape)

    @staticmethod
    def backward(ctx, g):
        a2, b2, y = ctx.saved_tensors
        g2 = _grad2(g, g.shape[-1])
        if ctx.op == 3:
            return None, g, g
        if ctx.op == 1:
            d = torch.empty_like(g2)
            nat.eltwise_f32(5, g2, y, d)              # g where y > 0
            return None, d.view(g.shape), None
        da, db = torch.empty_like(g2), torch.empty_like(g2)
        nat.eltwise_f32(0, g2, b2, da)
        nat.eltwise_f32(0, g2, a2, db)
        return None, da.view(g.shape), db.view(g.shape)


class ExpandBatchFn(torch.autograd.Function):
    """ViLBERT's `in_batch_pairs` / `fast_mode` batch expansion (mmf/models/vilbert.py:678-725) on fp32 activations [Bs, L, H] -> [reps * Bs, L, H]
    (mode 0: `unsqueeze(0).expand`, mode 1: `unsqueeze(1).expand`); backward = the sum over the broadcast index in a fixed order."""

    @staticmethod
    def forward(ctx, x, reps, mode):
        Bs, L, H = x.shape
        x2 = _rows(x)
        out = torch.empty(reps * Bs * L, H, dtype=F32, device=x2.device)
        nat.expand_batch(x2, out, Bs, reps, L * H, mode)
        ctx.meta = (Bs, L, H, reps, mode)
        return out.view(reps * Bs, L, H)

    @staticmethod
    def backward(ctx, g):
        Bs, L, H, reps, mode = ctx.meta
        g2 = _grad2(g, H)
        dx = torch.empty(Bs * L, H, dtype=F32, device=g2.device)
        nat.reduce_batch(g2, dx, Bs, reps, L * H, mode)
        return dx.view(Bs, L, H), None, None


def expand_batch(x, reps, mode):
    return ExpandBatchFn.apply(x, reps, mode)


# ---- MMBT (mmf/models/mmbt.py) and the MMF Transformer backend (mmf/models/transformers/backends/huggingface.py) ------------------------------
class MMBTEmbeddingsFn(torch.autograd.Function):
    """ModalEmbeddings.forward (mmbt.py:84-129) + the text BertEmbeddings (hf_layers.py:108-135), modal block first (mmbt.py:225): one fp32
    buffer [start token | N projected features | end token | T text], one LayerNorm, dropout (fp32_path.mmbt_embeddings is the forward).
    Backward: dropout, LayerNorm, then the rows go back where they came from — start / end / text rows into the word, position and type
    tables, the modal rows into the projection's weight gradient, its bias, the position rows s0 .. s0 + N - 1 and the modal type row."""

    @staticmethod
    def forward(ctx, feats, input_ids, start_tok, end_tok, text_type_ids, modal_type, word, pos, typ, ln_w, ln_b, proj_w, proj_b, eps, drop, pad_idx):
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
        mt_st, mt_en, coladd, rowtab, rowidx, mtype_rows, feat_types = mmbt_modal_types(modal_type, B, L, N, s0, dev, pd, td)
        ids = input_ids.contiguous(); tt = text_type_ids.contiguous()
        st = None if start_tok is None else start_tok.reshape(B, 1).contiguous()
        en = None if end_tok is None else end_tok.reshape(B, 1).contiguous()
        if st is not None:
            nat.embed_text_f32_fwd(st, mt_st, wd, pd, td, y, B, 1, S, H, 0, 0)
        if en is not None:
            nat.embed_text_f32_fwd(en, mt_en, wd, pd, td, y, B, 1, S, H, s0 + N, s0 + N)
        nat.embed_text_f32_fwd(ids, tt, wd, pd, td, y, B, T, S, H, L, 0)
        if D % 4:
            raise ValueError("fp32 path: modal feature width (%d) must be a multiple of 4" % D)
        f2 = feats.reshape(B * N, D)
        f2 = (f2 if f2.dtype == F32 else f2.float()).contiguous()
        posidx = (torch.arange(N, device=dev, dtype=torch.int64) + s0).repeat(B)
        nat.gemm_f32(f2, _w(proj_w), y, B * N, H, D, D, D, H, bias=_w(proj_b), coladd=coladd, rowtab=rowtab, rowidx=rowidx,
                     rowtab_ld=H, grp=(N, S - N, s0))
        # the type ids of the start token rows, the end token rows and the N feature rows of every sample (one id repeated, or a caller's own)
        if feat_types is None:
            feat_types = mtype_rows[:, s0:s0 + N].reshape(-1).contiguous()
        mt = torch.stack([mt_st.reshape(-1), mt_en.reshape(-1)]).contiguous()
        out, mean, rstd = _ln_fwd(y, ln_w, ln_b, eps)
        if drop[1]:
            o2 = torch.empty_like(out)
            nat.dropout_f32(out, o2, drop)
            out = o2
        ctx.save_for_backward(ids, tt, st, en, mt, f2, posidx, y, mean, rstd, ln_w.detach(), feat_types)
        ctx.meta = (B, N, T, H, s0, L, S, drop, pad_idx, word.shape[0], pos.shape[0], typ.shape[0])
        return out.view(B, S, H)

    @staticmethod
    def backward(ctx, g):
        ids, tt, st, en, mt, f2, posidx, y, mean, rstd, ln_w, feat_types = ctx.saved_tensors
        B, N, T, H, s0, L, S, drop, pad_idx, V, NP, NT = ctx.meta
        dev = y.device
        dy, dln_w, dln_b = _ln_bwd(_drop_bwd(_grad2(g, H), drop), y, mean, rstd, ln_w)
        dword = torch.zeros(V, H, dtype=F32, device=dev); dpos = torch.zeros(NP, H, dtype=F32, device=dev); dtyp = torch.zeros(NT, H, dtype=F32, device=dev)
        skip = -1 if pad_idx is None else int(pad_idx)

        def rows(idx, tab, n, off, sk=-1):         # rows off .. off + n - 1 of every sample's block scattered into `tab` by idx [B * n]
            nat.scatter_add_rows_f32(dy, H, B * n, H, idx, tab, H, grp=(n, S, off), skip=sk)
        for tok, off, mtb in ((st, 0, mt[0]), (en, s0 + N, mt[1])):
            if tok is not None:
                rows(tok.reshape(-1), dword, 1, off, skip)
                rows(torch.full((B,), off, dtype=torch.int64, device=dev), dpos, 1, off)
                rows(mtb.contiguous(), dtyp, 1, off)
        rows(ids.reshape(-1), dword, T, L, skip)
        rows(torch.arange(T, device=dev).repeat(B), dpos, T, L)
        rows(tt.reshape(-1), dtyp, T, L)
        # the modal rows: projection weight / bias gradients, their position rows, the modal type row
        dmod = torch.empty(B * N, H, dtype=F32, device=dev)
        nat.copy_rows(dy.view(torch.bfloat16)[s0:], S, dmod.view(torch.bfloat16), N, B, N, 2 * H)
        dpw, dpb = _wgrad(dmod, f2), _colsum(dmod)
        nat.scatter_add_rows_f32(dmod, H, B * N, H, posidx, dpos, H)
        nat.scatter_add_rows_f32(dmod, H, B * N, H, feat_types, dtyp, H)       # every feature row into the type row of its position
        return None, None, None, None, None, None, dword, dpos, dtyp, dln_w, dln_b, dpw, dpb, None, None, None


class AddPosTypeFn(torch.autograd.Function):
    """total = tok + pos_emb(arange(L)) + token_type_embeddings(segment_ids) (huggingface.py:147-155) on fp32 rows."""

    @staticmethod
    def forward(ctx, x, seg, pos, typ):
        B, L, H = x.shape
        sg = seg.contiguous() if (seg is not None and typ is not None) else None
        y = P.add_pos_type(x, sg, pos, typ if sg is not None else None)
        ctx.save_for_backward(sg)
        ctx.meta = (B, L, H, None if pos is None else pos.shape[0], None if (typ is None or sg is None) else typ.shape[0])
        return y

    @staticmethod
    def backward(ctx, g):
        (sg,) = ctx.saved_tensors
        B, L, H, NP, NT = ctx.meta
        g2 = _grad2(g, H)
        dpos = dtyp = None
        if NP is not None:
            dpos = torch.zeros(NP, H, dtype=F32, device=g2.device)
            nat.scatter_add_rows_f32(g2, H, B * L, H, torch.arange(L, device=g2.device).repeat(B), dpos, H)
        if NT is not None:
            dtyp = torch.zeros(NT, H, dtype=F32, device=g2.device)
            nat.scatter_add_rows_f32(g2, H, B * L, H, sg.reshape(-1), dtyp, H)
        return g, None, dpos, dtyp


class ConcatRowsFn(torch.autograd.Function):
    """torch.cat(list_embeddings, dim=1) (huggingface.py:159) of fp32 [B, L_m, H] blocks; backward = the row copies reversed."""

    @staticmethod
    def forward(ctx, *xs):
        ctx.lens = [int(x.shape[1]) for x in xs]
        return P.concat_rows(*xs)

    @staticmethod
    def backward(ctx, g):
        B, S, H = g.shape
        g2 = _grad2(g, H).view(torch.bfloat16)            # [B * S, 2H] 16-bit words
        outs, off = [], 0
        for L in ctx.lens:
            d = torch.empty(B * L, H, dtype=F32, device=g.device)
            nat.copy_rows(g2[off:], S, d.view(torch.bfloat16), L, B, L, 2 * H)
            outs.append(d.view(B, L, H))
            off += L
        return tuple(outs)


# ---- UNITER (mmf/models/uniter.py) -------------------------------------------------------------------------------------------------------
class FeatureTableAddFn(torch.autograd.Function):
    """img_feat + mask_embedding(img_masks) (uniter.py:74-78) on fp32 rows; the features get no gradient, the table's row `padding_idx` neither."""

    @staticmethod
    def forward(ctx, feats, idx, table, padding_idx):
        y = P.feature_table_add(feats, idx, table)
        ix = None if idx is None else idx.reshape(-1).long().contiguous()
        ctx.save_for_backward(ix)
        ctx.meta = (table.shape[0], padding_idx)
        return y

    @staticmethod
    def backward(ctx, g):
        (ix,) = ctx.saved_tensors
        NT, pad = ctx.meta
        if ix is None:
            return None, None, None, None
        D = g.shape[-1]
        g2 = _grad2(g, D)
        dt = torch.zeros(NT, D, dtype=F32, device=g2.device)
        nat.scatter_add_rows_f32(g2, D, g2.shape[0], D, ix, dt, D, skip=-1 if pad is None else int(pad))
        return None, None, dt, None


class SmallKLinearFn(torch.autograd.Function):
    """nn.Linear over a handful of input features (UNITER's 7-d box geometry, uniter.py:64,81), operand zero-padded to 16-byte rows; the input
    gets no gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        K = x.shape[-1]
        KP = (K + 3) // 4 * 4
        x2 = x.reshape(-1, K)
        x2 = (x2 if x2.dtype == F32 else x2.float()).contiguous()
        w = _w(weight)
        if KP != K:
            x2, w = P._pad_k(x2, K, KP), P._pad_k(w, K, KP)
        y = _gemm(x2, w, w.shape[0], bias=_w(bias))
        ctx.save_for_backward(x2)
        ctx.meta = (x.shape, K)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, g):
        (x2,) = ctx.saved_tensors
        xshape, K = ctx.meta
        dz = _pad4(_grad2(g, g.shape[-1]))
        return None, _wgrad(dz, x2)[:, :K].contiguous(), _colsum(dz)


class MaskedLMHeadFn(torch.autograd.Function):
    """Decoder tied to the word embeddings + CrossEntropyLoss(ignore_index) (visual_bert.py:267-277; HF BertLMPredictionHead) in fp32:
    (loss, logits [*, vocab]); only the loss carries gradient.  The backward materialises the fp32 [rows, vocab] gradient (the throughput path
    writes a bf16 GEMM operand instead) and runs the decoder's dgrad / weight gradient on it."""

    @staticmethod
    def forward(ctx, x, weight, bias, labels, ignore_index):
        x2 = _rows(x)
        M = x2.shape[0]
        w = _w(weight)
        N = w.shape[0]
        logits = _gemm(x2, w, N, bias=_w(bias))
        lab = labels.reshape(M).contiguous().long()
        lse = _empty(M, like=x2); rowloss = _empty(M, like=x2); loss = _empty(1, like=x2); count = _empty(1, like=x2)
        nat.vocab_cross_entropy_fwd(logits, lab, lse, rowloss, loss, count, M, N, ignore_index)
        ctx.save_for_backward(x2, w, logits, lab, lse, count)
        ctx.meta = (x.shape, ignore_index)
        ctx.mark_non_differentiable(logits)
        return loss[0], logits.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g, _glogits):
        x2, w, logits, lab, lse, count = ctx.saved_tensors
        xshape, ignore_index = ctx.meta
        M, N = logits.shape
        NP = (N + 3) // 4 * 4
        dl = _empty(M, NP, like=x2)
        nat.vocab_cross_entropy_f32_bwd(logits, lab, lse, count, g.float().reshape(1).contiguous(), dl, NP, M, N, ignore_index)
        dz = dl[:, :N]
        return _dgrad(dz, w).view(xshape), _wgrad(dz, x2), _colsum(dz), None, None


class MaskedMeanFn(torch.autograd.Function):
    """(x * mask.unsqueeze(-1)).sum(1) / mask.sum(1, keepdim=True): the text pooling of ViLBERT's dynamic_attention (vilbert.py:204-205) on
    fp32 rows.  x [B, T, H], mask [B, T] (0 / 1) -> [B, H]."""

    @staticmethod
    def forward(ctx, x, mask):
        B, T, H = x.shape
        m = mask.detach().reshape(B, T).float().contiguous()
        pool = _empty(B, H, like=x)
        nat.masked_mean_f32(_rows(x), m, pool, B, T, H)
        ctx.save_for_backward(m)
        ctx.meta = (B, T, H)
        return pool

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        B, T, H = ctx.meta
        dx = _empty(B * T, H, like=m)
        nat.masked_mean_f32_bwd(g.float().contiguous(), m, dx, B, T, H)
        return dx.view(B, T, H), None


class MaskedRegionHeadFn(torch.autograd.Function):
    """The decoder + loss of ViLBERT's masked-region classification (vilbert.py:846-858 BertImagePredictionHead.decoder, :1150-1157
    `visual_target: 0`) in fp32: KLDivLoss(log_softmax(h W^T + b), target) summed over the regions with image_label == 1, divided by their number.
    Returns (loss, scores); only the loss carries gradient.  Backward: the fp32 [rows, classes] gradient, then the decoder's dgrad / weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, target, row_label):
        x2 = _rows(x)
        M = x2.shape[0]
        w = _w(weight)
        N = w.shape[0]
        logits = _gemm(x2, w, N, bias=_w(bias))
        lab = row_label.reshape(M).long().contiguous()
        tgt = target.reshape(M, N).float().contiguous()
        lse = _empty(M, like=x2); tsum = _empty(M, like=x2); rowloss = _empty(M, like=x2); loss = _empty(1, like=x2); count = _empty(1, like=x2)
        nat.soft_target_kl_fwd(logits, tgt, lab, lse, tsum, rowloss, loss, count, M, N)
        ctx.save_for_backward(x2, w, logits, tgt, lab, lse, tsum, count)
        ctx.meta = (x.shape,)
        out = logits.view(*x.shape[:-1], N)
        ctx.mark_non_differentiable(out)
        return loss[0], out

    @staticmethod
    def backward(ctx, g, _gscores):
        x2, w, logits, tgt, lab, lse, tsum, count = ctx.saved_tensors
        (xshape,) = ctx.meta
        M, N = logits.shape
        NP = (N + 3) // 4 * 4
        dl = _empty(M, NP, like=x2)
        nat.soft_target_kl_f32_bwd(logits, tgt, lab, lse, tsum, count, g.float().reshape(1).contiguous(), dl, NP, M, N)
        dz = dl[:, :N]
        return _dgrad(dz, w).view(xshape), _wgrad(dz, x2), _colsum(dz), None, None


class MaskedRegionRegressionFn(torch.autograd.Function):
    """ViLBERT `visual_target: 1` (vilbert.py:1074-1075, 1139-1148) in fp32: decoder GEMM, nn.MSELoss(reduction="none") over the regions with
    image_label == 1 divided by max(their element count, 1).  Returns (loss, scores); only the loss carries gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, target, row_label):
        x2 = _rows(x)
        M = x2.shape[0]
        w = _w(weight)
        N = w.shape[0]
        pred = _gemm(x2, w, N, bias=_w(bias))
        tgt = target.reshape(M, N).float().contiguous()
        lab = row_label.reshape(M).long().contiguous()
        loss = _empty(1, like=x2); count = _empty(1, like=x2)
        nat.mse_fwd(pred, tgt, loss, M, N, row_label=lab, count=count)
        ctx.save_for_backward(x2, w, pred, tgt, lab, count)
        ctx.meta = (x.shape,)
        out = pred.view(*x.shape[:-1], N)
        ctx.mark_non_differentiable(out)
        return loss[0], out

    @staticmethod
    def backward(ctx, g, _gscores):
        x2, w, pred, tgt, lab, count = ctx.saved_tensors
        (xshape,) = ctx.meta
        M, N = pred.shape
        NP = (N + 3) // 4 * 4
        dl = _empty(M, NP, like=x2)
        nat.mse_f32_bwd(pred, tgt, g.float().reshape(1).contiguous(), dl, NP, M, N, row_label=lab, count=count)
        dz = dl[:, :N]
        return _dgrad(dz, w).view(xshape), _wgrad(dz, x2), _colsum(dz), None, None


class MaskedRegionNCEFn(torch.autograd.Function):
    """ViLBERT `visual_target: 2` (vilbert.py:1158-1227) in fp32: decoder GEMM, every masked region's prediction scored against its own target
    and K sampled negatives, CrossEntropyLoss against class 0.  `neg_index`: flat region indices [B, R, K] drawn on the host side of the boundary."""

    @staticmethod
    def forward(ctx, x, weight, bias, target, row_label, neg_index):
        x2 = _rows(x)
        M = x2.shape[0]
        w = _w(weight)
        N = w.shape[0]
        pred = _gemm(x2, w, N, bias=_w(bias))
        tgt = target.reshape(M, N).float().contiguous()
        lab = row_label.reshape(M).long().contiguous()
        neg = neg_index.reshape(M, -1).long().contiguous()
        NK = neg.shape[1]
        scores = _empty(M, NK + 1, like=x2)
        lse = _empty(M, like=x2); rowloss = _empty(M, like=x2); loss = _empty(1, like=x2); count = _empty(1, like=x2)
        nat.nce_fwd(pred, tgt, neg, lab, scores, lse, rowloss, loss, count, M, N, NK)
        ctx.save_for_backward(x2, w, tgt, neg, lab, scores, lse, count)
        ctx.meta = (x.shape, N, NK)
        out = pred.view(*x.shape[:-1], N)
        ctx.mark_non_differentiable(out)
        return loss[0], out

    @staticmethod
    def backward(ctx, g, _gscores):
        x2, w, tgt, neg, lab, scores, lse, count = ctx.saved_tensors
        xshape, N, NK = ctx.meta
        M = x2.shape[0]
        NP = (N + 3) // 4 * 4
        dl = _empty(M, NP, like=x2)
        nat.nce_f32_bwd(tgt, neg, lab, scores, lse, count, g.float().reshape(1).contiguous(), dl, NP, M, N, NK)
        dz = dl[:, :N]
        return _dgrad(dz, w).view(xshape), _wgrad(dz, x2), _colsum(dz), None, None, None


# ---- M4C's stages (mmf/models/m4c.py:185-304) in fp32, forward and backward ------------------------------------------------------------
class L2NormRowsFn(torch.autograd.Function):
    """F.normalize(x, dim=-1) (m4c.py:195) on fp32 rows; backward dx = (g - y <g, y>) / max(||x||, eps)."""

    @staticmethod
    def forward(ctx, x):
        D = x.shape[-1]
        x2 = _rows(x)
        y = torch.empty_like(x2)
        nat.l2norm_rows_f32(x2, D, y, D, x2.shape[0], D)
        ctx.save_for_backward(x2, y)
        ctx.xshape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        x2, y = ctx.saved_tensors
        rows, D = y.shape
        dx = torch.empty_like(y)
        nat.l2norm_rows_f32_bwd(_grad2(g, D), D, y, D, x2, D, dx, D, rows, D)
        return dx.view(ctx.xshape)


class OcrFeatureConcatFn(torch.autograd.Function):
    """cat([normalize(fasttext), normalize(phoc), normalize(fc7), zeros(order vectors)], -1) (m4c.py:211-237) as fp32 rows padded to a multiple
    of 4 columns; only the appearance feature (`fc7`, an activation) carries a gradient."""

    @staticmethod
    def forward(ctx, fasttext, phoc, fc7, order_dim):
        B, N, _ = fasttext.shape
        rows = B * N
        d0, d1, d2 = fasttext.shape[-1], phoc.shape[-1], fc7.shape[-1]
        K = d0 + d1 + d2 + int(order_dim)
        KP = (K + 3) // 4 * 4
        out = torch.zeros(rows, KP, dtype=F32, device=fc7.device)
        f0, f1, f2 = (t.reshape(rows, d).float().contiguous() for t, d in ((fasttext, d0), (phoc, d1), (fc7.detach(), d2)))
        nat.l2norm_rows_f32(f0, d0, out, KP, rows, d0)
        nat.l2norm_rows_f32(f1, d1, out[:, d0:], KP, rows, d1)
        nat.l2norm_rows_f32(f2, d2, out[:, d0 + d1:], KP, rows, d2)
        ctx.save_for_backward(out, f2)
        ctx.meta = (B, N, d0 + d1, d2, KP)
        return out.view(B, N, KP)

    @staticmethod
    def backward(ctx, g):
        out, f2 = ctx.saved_tensors
        B, N, off, d2, KP = ctx.meta
        rows = B * N
        g2 = _grad2(g, KP)
        dx = _empty(rows, d2, like=out)
        nat.l2norm_rows_f32_bwd(g2[:, off:], KP, out[:, off:], KP, f2, d2, dx, d2, rows, d2)
        return None, None, dx.view(B, N, d2), None


class PaddedLinearFn(torch.autograd.Function):
    """nn.Linear on rows already zero-padded to KP = round_up(K, 4) columns (the 3002-wide OCR feature, m4c.py:243): the weight is padded the
    same way for the GEMMs; its gradient is cut back to K columns."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        N, K = weight.shape
        KP = x.shape[-1]
        x2 = _rows(x)
        w = _w(weight)
        if KP != K:
            w = P._pad_k(w, K, KP)
        y = _gemm(x2, w, N, bias=_w(bias))
        ctx.save_for_backward(x2, w)
        ctx.meta = (x.shape, N, K)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, w = ctx.saved_tensors
        xshape, N, K = ctx.meta
        dz = _grad2(g, N)
        dx = _dgrad(dz, w).view(xshape) if ctx.needs_input_grad[0] else None
        return dx, _wgrad(dz, x2)[:, :K].contiguous(), _colsum(dz)


class PrevPredGatherFn(torch.autograd.Function):
    """_batch_gather(cat([ans_emb.expand(B), ocr_emb], 1), prev_inds) (m4c.py:526-528) as one two-source row gather; backward scatter-adds the
    row gradients into the two sources (repeated indices collide: fp32 atomics)."""

    @staticmethod
    def forward(ctx, ans, ocr, prev_inds):
        V, H = ans.shape
        B, N, _ = ocr.shape
        T = prev_inds.shape[1]
        batch = torch.arange(B, device=prev_inds.device, dtype=torch.int64).unsqueeze(1) * N
        flat = torch.where(prev_inds < V, prev_inds, prev_inds + batch).contiguous()     # OCR row (b, i) -> V + b N + i
        out = _empty(B * T, H, like=ocr)
        nat.gather_rows2_f32(_rows(ans), _rows(ocr), flat, out, B * T, H)
        ctx.save_for_backward(flat)
        ctx.meta = (V, B, N, T, H)
        return out.view(B, T, H)

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        V, B, N, T, H = ctx.meta
        g2 = _grad2(g, H)
        idx = flat.reshape(-1)
        dans = torch.zeros(V, H, dtype=F32, device=g2.device)
        docr = torch.zeros(B * N, H, dtype=F32, device=g2.device)
        nat.scatter_add_rows_f32(g2, H, B * T, H, idx, dans, H)                 # rows with idx >= V fall outside this table: dropped by the kernel
        nat.scatter_add_rows_f32(g2, H, B * T, H, idx - V, docr, H)             # rows with idx < V become negative: dropped
        return dans, docr.view(B, N, H), None


class SplitRowsFn(torch.autograd.Function):
    """The slices `mmt_seq_output[:, a:b]` of MMT.forward (m4c.py:446-449) as strided copies of fp32 rows (moved as pairs of 16-bit words);
    backward writes the block gradients back into one [B, S, H] buffer, blocks nobody used stay zero."""

    @staticmethod
    def forward(ctx, x, lens):
        B, S, H = x.shape
        lens = [int(l) for l in lens]
        assert sum(lens) == S
        xb = _rows(x).view(torch.bfloat16)
        outs, off = [], 0
        for L in lens:
            d = _empty(B * L, H, like=x)
            nat.copy_rows(xb[off:], S, d.view(torch.bfloat16), L, B, L, 2 * H)
            outs.append(d.view(B, L, H))
            off += L
        ctx.meta = (B, S, H, lens)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        B, S, H, lens = ctx.meta
        dev = next(g.device for g in gs if g is not None)
        dx = torch.zeros(B * S, H, dtype=F32, device=dev)
        dxb = dx.view(torch.bfloat16)
        off = 0
        for g, L in zip(gs, lens):
            if g is not None:
                nat.copy_rows(_grad2(g, H).view(torch.bfloat16), L, dxb[off:], S, B, L, 2 * H)
            off += L
        return dx.view(B, S, H), None


class M4CScoresFn(torch.autograd.Function):
    """M4C._forward_output (m4c.py:275-283) in fp32: `classifier(dec)` and the OCR pointer scores (:474-493) written by the two producers into
    one [B, T, V + N] buffer.  Backward: the classifier's dgrad / weight gradient on the fixed-vocabulary columns (copied out as 16-byte rows), the
    pointer scores' dq = ds k / sqrt(d), dk = ds^T q / sqrt(d), then the query / key projections' gradients."""

    @staticmethod
    def forward(ctx, dec, ocr, cls_w, cls_b, q_w, q_b, k_w, k_b, ocr_mask_add):
        B, T, H = dec.shape
        N = ocr.shape[1]
        V, HQ = cls_w.shape[0], q_w.shape[0]
        d2, o2 = _rows(dec), _rows(ocr)
        cw, qw, kw = _w(cls_w), _w(q_w), _w(k_w)
        out = _empty(B * T, V + N, like=d2)
        nat.gemm_f32(d2, cw, out, B * T, V, H, H, H, V + N, bias=_w(cls_b))
        q = _gemm(d2, qw, HQ, bias=_w(q_b))
        k = _gemm(o2, kw, HQ, bias=_w(k_b))
        scale = 1.0 / math.sqrt(HQ)
        nat.ptr_scores_f32(q, k, ocr_mask_add.reshape(B, N).float().contiguous(), out[:, V:], V + N, B, T, N, HQ, scale)
        ctx.save_for_backward(d2, o2, q, k, cw, qw, kw)
        ctx.meta = (B, T, N, H, V, HQ, scale)
        return out.view(B, T, V + N)

    @staticmethod
    def backward(ctx, g):
        d2, o2, q, k, cw, qw, kw = ctx.saved_tensors
        B, T, N, H, V, HQ, scale = ctx.meta
        M = B * T
        g2 = _grad2(g, V + N)
        VP = (V + 3) // 4 * 4
        dfix_p = _empty(M, VP, like=d2)
        nat.slice_rows_f32(g2, V + N, V, dfix_p, VP, M)
        dfix = dfix_p[:, :V]
        ddec = _dgrad(dfix, cw)
        dcw, dcb = _wgrad(dfix, d2), _colsum(dfix)
        dq = _empty(M, HQ, like=d2); dk = _empty(B * N, HQ, like=d2)
        nat.ptr_scores_f32_bwd(g2[:, V:], V + N, q, k, dq, dk, B, T, N, HQ, scale)
        ddec = _dgrad(dq, qw, resid=ddec)
        dqw, dqb = _wgrad(dq, d2), _colsum(dq)
        docr = _dgrad(dk, kw)
        dkw, dkb = _wgrad(dk, o2), _colsum(dk)
        return ddec.view(B, T, H), docr.view(B, N, H), dcw, dcb, dqw, dqb, dkw, dkb, None


class LogitBCEFn(torch.autograd.Function):
    """mean(BCEWithLogits(scores, targets)) * num_labels (losses.py:246-251) with an fp32 gradient."""

    @staticmethod
    def forward(ctx, scores, targets):
        B, N = scores.shape
        s = scores.float().contiguous()
        t = targets.float().contiguous()
        loss = torch.empty(1, dtype=F32, device=s.device)
        nat.bce_logits_fwd(s, t, loss, B, N)
        ctx.save_for_backward(s, t)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        s, t = ctx.saved_tensors
        d = torch.empty_like(s)
        nat.bce_logits_f32_bwd(s, t, g.float().reshape(1).contiguous(), d, s.shape[0], s.shape[1])
        return d, None


# ---- the operator surface (called from mmf_amd/ops.py while `active()`) ---------------------------------------------------------------
def visio_linguistic_embeddings(input_ids, token_type_ids, feats, vtype, word, pos, typ, ln_w, ln_b, typ_vis, pos_vis, proj_w, proj_b, eps, p,
                                training, pad_idx, image_text_alignment=None):
    return VisioLinguisticEmbeddingsFn.apply(input_ids, token_type_ids, feats, vtype, word, pos, typ, ln_w, ln_b, typ_vis, pos_vis, proj_w, proj_b,
                                             eps, make_drop(p, training), pad_idx if pad_idx is not None and pad_idx >= 0 else None,
                                             image_text_alignment)


def transformer_layer(x, wq, bq, wk, bk, wv, bv, wo, bo, ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b, mask_add, heads, eps1, eps2, p_attn, p_hid1,
                      p_hid2, training, causal_tail):
    a = AttentionBlockFn.apply(x, wq, bq, wk, bk, wv, bv, wo, bo, ln1_w, ln1_b, mask_add, heads, eps1, make_drop(p_attn, training),
                               make_drop(p_hid1, training), causal_tail)
    return FeedForwardFn.apply(a, w1, b1, w2, b2, ln2_w, ln2_b, eps2, make_drop(p_hid2, training))


def linear(x, weight, bias):
    return LinearFn.apply(x, weight, bias, 0)


def dense_gelu(x, weight, bias):
    return LinearFn.apply(x, weight, bias, 1)


def linear_tanh(x, weight, bias):
    return LinearFn.apply(x, weight, bias, 3)


def layer_norm(x, gamma, beta, eps):
    return LayerNormFn.apply(x, gamma, beta, eps)


def dropout(x, p, training):
    drop = make_drop(p, training)
    return DropoutFn.apply(x, drop) if drop[1] else x


def gather_rows(x, index, p, training):
    return dropout(GatherRowsFn.apply(x, index), p, training)


def logit_bce(scores, targets):
    return LogitBCEFn.apply(scores, targets)


def pair_halves(x):
    return PairHalvesFn.apply(x)


def dense_residual_ln(h, resid, weight, bias, gamma, beta, eps, p, training):
    return DenseResidualLNFn.apply(h, resid, weight, bias, gamma, beta, eps, make_drop(p, training))


def feed_forward(x, w1, b1, w2, b2, gamma, beta, eps, p, training):
    return FeedForwardFn.apply(x, w1, b1, w2, b2, gamma, beta, eps, make_drop(p, training))


def bi_attention(img, txt, q1, k1, v1, q2, k2, v2, img_mask_add, txt_mask_add, heads, p1, p2, training):
    return BiAttentionFn.apply(img, txt, q1.weight, q1.bias, k1.weight, k1.bias, v1.weight, v1.bias, q2.weight, q2.bias, k2.weight, k2.bias,
                               v2.weight, v2.bias, img_mask_add, txt_mask_add, heads, make_drop(p1, training), make_drop(p2, training))


def image_feature_embeddings(feats, loc, w_img, b_img, w_loc, b_loc, ln_w, ln_b, eps, p, training):
    return ImageFeatureEmbeddingsFn.apply(feats, loc, w_img, b_img, w_loc, b_loc, ln_w, ln_b, eps, make_drop(p, training))


def eltwise_mul(a, b):
    return EltwiseFn.apply(0, a, b)


def relu(a):
    return EltwiseFn.apply(1, a, None)


def add(a, b):
    return EltwiseFn.apply(3, a, b)


def mmbt_embeddings(feats, input_ids, start_tok, end_tok, text_type_ids, modal_type, word, pos, typ, ln_w, ln_b, proj_w, proj_b, eps, p, training,
                    pad_idx):
    return MMBTEmbeddingsFn.apply(feats, input_ids, start_tok, end_tok, text_type_ids, modal_type, word, pos, typ, ln_w, ln_b, proj_w, proj_b, eps,
                                  make_drop(p, training), pad_idx)


def add_pos_type(x, seg, pos, typ):
    return AddPosTypeFn.apply(x, seg, pos, typ)


def concat_rows(*xs):
    return ConcatRowsFn.apply(*xs)


def feature_table_add(feats, idx, table, padding_idx=0):
    return FeatureTableAddFn.apply(feats, idx, table, padding_idx)


def small_k_linear(x, weight, bias):
    return SmallKLinearFn.apply(x, weight, bias)


def masked_lm_head(x, weight, bias, labels, ignore_index):
    return MaskedLMHeadFn.apply(x, weight, bias, labels, ignore_index)


def masked_region_head(x, weight, bias, target, row_label):
    return MaskedRegionHeadFn.apply(x, weight, bias, target, row_label)


def masked_region_regression(x, weight, bias, target, row_label):
    return MaskedRegionRegressionFn.apply(x, weight, bias, target, row_label)


def masked_region_nce(x, weight, bias, target, row_label, neg_index):
    return MaskedRegionNCEFn.apply(x, weight, bias, target, row_label, neg_index)


def masked_mean(x, mask):
    return MaskedMeanFn.apply(x, mask)


def attention_block(x, wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, mask_add, heads, eps, p_attn, p_hid, training, qk_gate=None, causal_tail=0):
    return AttentionBlockFn.apply(x, wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, mask_add, heads, eps, make_drop(p_attn, training),
                                  make_drop(p_hid, training), causal_tail, qk_gate)


def l2norm_rows(x):
    return L2NormRowsFn.apply(x)


def ocr_feature_concat(fasttext, phoc, fc7, order_dim):
    return OcrFeatureConcatFn.apply(fasttext, phoc, fc7, order_dim)


def padded_linear(x, weight, bias):
    return PaddedLinearFn.apply(x, weight, bias)


def prev_pred_gather(ans, ocr, prev_inds):
    return PrevPredGatherFn.apply(ans, ocr, prev_inds)


def split_rows(x, lens):
    return SplitRowsFn.apply(x, lens)


def m4c_scores(dec, ocr, cls_w, cls_b, q_w, q_b, k_w, k_b, ocr_mask_add):
    return M4CScoresFn.apply(dec, ocr, cls_w, cls_b, q_w, q_b, k_w, k_b, ocr_mask_add)


def unsupported(name):
    raise NotImplementedError("mmf_amd.fp32_training(): operator `%s` has no fp32 backward yet (built: the VisualBERT classification step)" % name)
