"""CPU stand-in for mmf_amd._native used by the host-logic ("dry run") tests: every kernel entry point becomes a
shape / dtype / extent CHECK with no arithmetic, so the Python side of a model — autograd Functions, buffer sizes,
leading dimensions, argument order, the number of gradients each backward returns — can be exercised here, where there
is no GPU.  It computes nothing: outputs keep whatever torch.empty gave them.  Numerical parity is the `-m gpu` tests' job."""
import contextlib

import torch

from mmf_amd import _native as N

_INT_RETURNS = {
    "layernorm_bwd_ws_floats": lambda H: 64 * 3 * H, "colsum_ws_floats": lambda n: 64 * n,
    "gemm_rowsum_supported": lambda M, Nn, K: True,
    "layernorm_dropout_fusable": lambda H: H % 256 == 0 and H <= 1024,
    "attention_keep_bits_words": lambda B, heads, Sq, Sk, head_dim=64: (B * heads * ((Sq + 31) // 32) * ((Sk + 31) // 32) * 32
                                                                        if max(Sq, Sk) <= (256 if head_dim == 64 else 128) else 0),
}
_KEEP = {"drop_cfg", "_drop4", "lib", "_check", "_stream", "_p", "_req", "gemm_site"}
calls = []


def _room(t):
    """elements addressable from t's first element to the end of its storage"""
    return t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()


def _need(t, rows, ld, cols, what):
    if t is None:
        return
    need = (rows - 1) * ld + cols
    assert _room(t) >= need, "%s: needs %d elements from its base pointer, buffer has %d" % (what, need, _room(t))


def _gemm(A, B, C_out, M, Nn, K, lda, ldb, ldc, a_kmajor=False, b_kmajor=False, beta=0.0, bias=None, coladd=None, rowtab=None,
          rowidx=None, rowtab_ld=0, act=0, U=None, aux=None, resid=None, ldr=0, drop=N.NO_DROP, grp=(0, 0, 0), debug_flags=0,
          rowsum_out=None):
    assert A.dtype in (torch.bfloat16, torch.float32) and B.dtype in (torch.bfloat16, torch.float32)
    assert lda % 8 == 0 and ldb % 8 == 0, "gemm: lda/ldb must be multiples of 8 (%d, %d)" % (lda, ldb)
    k8 = (K + 7) // 8 * 8
    if a_kmajor:
        _need(A, K, lda, (M + 7) // 8 * 8 if lda >= (M + 7) // 8 * 8 else M, "gemm A (k-major)")
    else:
        assert lda >= k8, "gemm: lda %d does not cover round_up(K=%d, 8)" % (lda, K)
        _need(A, M, lda, K, "gemm A")
    if b_kmajor:
        _need(B, K, ldb, Nn, "gemm B (k-major)")
    else:
        assert ldb >= k8, "gemm: ldb %d does not cover round_up(K=%d, 8)" % (ldb, K)
        _need(B, Nn, ldb, K, "gemm B")
    rows_out = M if grp[0] == 0 else (M - 1) + ((M - 1) // grp[0]) * grp[1] + grp[2] + 1
    _need(C_out, rows_out, ldc, Nn, "gemm C")
    for v, n in ((bias, Nn), (coladd, Nn)):
        if v is not None:
            assert v.dtype == torch.float32 and v.numel() >= n
    if resid is not None:
        _need(resid, M, ldr, Nn, "gemm resid")
    if rowsum_out is not None:
        assert rowsum_out.numel() >= M
    calls.append(("gemm", M, Nn, K))


def _gemm_grouped(problems):
    assert 1 <= len(problems) <= N.GEMM_GROUP_MAX
    key = None
    for kw in problems:
        kw = dict(kw)
        k = (bool(kw.get("a_kmajor")), bool(kw.get("b_kmajor")), kw["A"].dtype, kw["B"].dtype)
        assert key is None or key == k, "gemm_grouped: mixed operand layouts"
        key = k
        _gemm(kw.pop("A"), kw.pop("B"), kw.pop("C_out"), kw.pop("M"), kw.pop("N"), kw.pop("K"), kw.pop("lda"), kw.pop("ldb"),
              kw.pop("ldc"), **kw)


def _attention_fwd(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, lse, B, heads, Sq, Sk, scale, drop=N.NO_DROP, head_dim=64, ctx_f32=None,
                   causal_tail=0, q_batch_rows=0, kv_batch_rows=0, mask_batch_stride=0, keep_bits=None, keep_lanes=None):
    if keep_lanes is not None:     # decisions drawn ahead (mmf_attn_desc.keep_lanes): the shapes that take keep_bits, dropout on
        assert keep_lanes.dtype == torch.int32 and drop[1] and max(Sq, Sk) <= (256 if head_dim == 64 else 128)
        assert keep_lanes.numel() == B * heads * ((Sq + 31) // 32) * 256
    assert head_dim in (64, 128) and 0 <= causal_tail <= Sk and (causal_tail == 0 or (Sq == Sk and head_dim == 64))
    if keep_bits is not None:      # the dropout keep-bit table (mmf_attn_desc.keep_bits): only where the kernels take one, only with dropout on
        assert keep_bits.dtype == torch.int32 and drop[1] and max(Sq, Sk) <= (256 if head_dim == 64 else 128)
        assert keep_bits.numel() == B * heads * ((Sq + 31) // 32) * ((Sk + 31) // 32) * 32 and not q_batch_rows and not kv_batch_rows
    assert Sq <= 256 and Sk <= 256
    qb, kb, mb = q_batch_rows or Sq, kv_batch_rows or Sk, mask_batch_stride or Sk
    assert qb >= Sq and kb >= Sk and mb >= Sk
    _need(q, (B - 1) * qb + Sq, ldq, heads * head_dim, "attention q"); _need(k, (B - 1) * kb + Sk, ldk, heads * head_dim, "attention k")
    _need(v, (B - 1) * kb + Sk, ldv, heads * head_dim, "attention v"); _need(ctx, B * Sq, ldo, heads * head_dim, "attention ctx")
    if mask is not None and mask.dim() == 3:       # a materialised per-query mask (mmf_attn_desc.mask_query_stride): head_dim 64, no causal tail beside it
        assert mask.dtype == torch.float32 and tuple(mask.shape) == (B, Sq, Sk) and mask.stride(2) == 1
        assert head_dim == 64 and causal_tail == 0 and not mask_batch_stride
    elif mask is not None and mask.dim() == 4:     # one [Sq, Sk] mask per head (mmf_attn_desc.mask_head_stride)
        assert mask.dtype == torch.float32 and tuple(mask.shape) == (B, heads, Sq, Sk) and mask.stride(3) == 1 and head_dim == 64 and causal_tail == 0
    elif mask is not None:
        assert mask.dtype == torch.float32 and mask.numel() >= (B - 1) * mb + Sk
    assert lse.numel() >= B * heads * Sq
    calls.append(("attention_fwd", B, heads, Sq, Sk, causal_tail) + (("per-query mask",) if (mask is not None and mask.dim() == 3) else ()) +
                 (("per-head mask",) if (mask is not None and mask.dim() == 4) else ()))


def _attention_bwd(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, lse, B, heads, Sq, Sk, scale, dctx, dq, dk, dv, delta, drop=N.NO_DROP,
                   head_dim=64, ctx_f32=None, causal_tail=0, keep_bits=None):
    _attention_fwd(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, lse, B, heads, Sq, Sk, scale, drop, head_dim, ctx_f32, causal_tail, keep_bits=keep_bits)
    _need(dctx, B * Sq, ldo, heads * head_dim, "attention dctx"); _need(dq, B * Sq, ldq, heads * head_dim, "attention dq")
    _need(dk, B * Sk, ldk, heads * head_dim, "attention dk"); _need(dv, B * Sk, ldv, heads * head_dim, "attention dv")
    calls[-1] = ("attention_bwd",) + calls[-1][1:]


def _attention_f32_bwd(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, lse, B, heads, Sq, Sk, scale, dctx, dq, dk, dv, delta, head_dim=64, causal_tail=0,
                       drop=None):
    hd = head_dim
    for t in (q, k, v, ctx, dctx, dq, dk, dv, lse, delta):
        assert t.dtype == torch.float32
    _need(q, B * Sq, ldq, heads * hd, "attention_f32_bwd q"); _need(dq, B * Sq, ldq, heads * hd, "attention_f32_bwd dq")
    _need(k, B * Sk, ldk, heads * hd, "attention_f32_bwd k"); _need(dk, B * Sk, ldk, heads * hd, "attention_f32_bwd dk")
    _need(v, B * Sk, ldv, heads * hd, "attention_f32_bwd v"); _need(dv, B * Sk, ldv, heads * hd, "attention_f32_bwd dv")
    _need(ctx, B * Sq, ldo, heads * hd, "attention_f32_bwd ctx"); _need(dctx, B * Sq, ldo, heads * hd, "attention_f32_bwd dctx")
    assert lse.numel() == B * heads * Sq == delta.numel()
    calls.append(("attention_f32_bwd", B, heads, Sq, Sk))


def _gemm_f32(A, B, C_out, M, Nn, K, lda, ldb, ldc, bias=None, coladd=None, rowtab=None, rowidx=None, rowtab_ld=0, act=0, resid=None,
              ldr=0, grp=(0, 0, 0), a_kmajor=False, b_kmajor=False, U=None, aux=None, drop=None, beta=0.0, split_k=False):
    for t in (A, B, C_out, resid, bias, coladd, rowtab, U, aux):
        assert t is None or t.dtype == torch.float32
    assert lda % 4 == 0 and ldb % 4 == 0 and ldc >= Nn and act in (0, 1, 2, 3, 4)
    assert not a_kmajor or b_kmajor
    assert A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0
    if a_kmajor:
        assert lda >= M
        _need(A, K, lda, M, "gemm_f32 A (k-major)")
    else:
        assert lda >= (K + 3) // 4 * 4
        _need(A, M, lda, K, "gemm_f32 A")
    if b_kmajor:
        assert ldb >= Nn
        _need(B, K, ldb, Nn, "gemm_f32 B (k-major)")
    else:
        assert ldb >= (K + 3) // 4 * 4
        _need(B, Nn, ldb, K, "gemm_f32 B")
    assert (act not in (2, 4)) or aux is not None
    assert U is None or act == 1
    for t in (U, aux):
        if t is not None:
            _need(t, M, ldc, Nn, "gemm_f32 U / aux")
    rows_out = M if grp[0] == 0 else (M - 1) + ((M - 1) // grp[0]) * grp[1] + grp[2] + 1
    _need(C_out, rows_out, ldc, Nn, "gemm_f32 C")
    for v in (bias, coladd):
        assert v is None or v.numel() >= Nn
    if rowtab is not None:
        assert rowidx.dtype == torch.int64 and rowidx.numel() >= M and rowtab_ld >= Nn
        assert int(rowidx.min()) >= 0 and int(rowidx.max()) < rowtab.shape[0]
    if resid is not None:
        assert ldr >= Nn
        _need(resid, M, ldr, Nn, "gemm_f32 resid")
    calls.append(("gemm_f32", M, Nn, K))


def _attention_f32_fwd(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, B, heads, Sq, Sk, scale, head_dim=64, causal_tail=0, lse=None, drop=None):
    assert lse is None or (lse.dtype == torch.float32 and lse.numel() == B * heads * Sq)
    assert (head_dim == 64 and Sk <= 256) or (head_dim == 128 and Sk <= 128)
    assert 0 <= causal_tail <= Sk and (causal_tail == 0 or Sq == Sk)
    for t in (q, k, v, ctx):
        assert t.dtype == torch.float32
    hd = head_dim
    _need(q, B * Sq, ldq, heads * hd, "attention_f32 q"); _need(k, B * Sk, ldk, heads * hd, "attention_f32 k")
    _need(v, B * Sk, ldv, heads * hd, "attention_f32 v"); _need(ctx, B * Sq, ldo, heads * hd, "attention_f32 ctx")
    assert mask is None or (mask.dtype == torch.float32 and mask.numel() >= B * Sk)
    calls.append(("attention_f32_fwd", B, heads, Sq, Sk))


def _layernorm_f32_fwd(x, gamma, beta, y, rows, H, eps):
    assert H % 4 == 0 and H <= 2048 and gamma.numel() == H == beta.numel()
    _need(x, rows, H, H, "layernorm_f32 x"); _need(y, rows, H, H, "layernorm_f32 y")
    calls.append(("layernorm_f32_fwd", rows, H))


def _embed_text_f32(ids, seg, word, pos, typ, y, B, T, S, H, row0=0, pos0=0):
    assert y.dtype == torch.float32 and ids.numel() == B * T and pos0 + T <= pos.shape[0]
    assert int(ids.max()) < word.shape[0] and int(seg.max()) < typ.shape[0]
    _need(y, (B - 1) * S + row0 + T, H, H, "embed_text_f32 y")


def _gather_rows_f32(x, index, out, B, S, H):
    assert x.dtype == torch.float32 and out.dtype == torch.float32 and index.numel() == B
    _need(x, B * S, H, H, "gather_rows_f32 x"); _need(out, B, H, H, "gather_rows_f32 out")


def _vocab_ce_fwd(logits, labels, lse, rowloss, loss, count, R, Cn, ignore_index=-1):
    assert logits.dtype == torch.float32 and logits.stride(0) >= Cn and labels.dtype == torch.int64 and labels.numel() == R
    _need(logits, R, logits.stride(0), Cn, "vocab_ce logits")
    assert lse.numel() >= R and rowloss.numel() >= R and loss.numel() == 1 and count.numel() == 1
    ok = labels[labels != ignore_index]
    assert ok.numel() == 0 or (int(ok.min()) >= 0 and int(ok.max()) < Cn), "label outside the vocabulary"
    calls.append(("vocab_cross_entropy_fwd", R, Cn))


def _vocab_ce_bwd(logits, labels, lse, count, gloss, dlogits, ldd, R, Cn, ignore_index=-1):
    assert ldd % 8 == 0 and ldd >= Cn and dlogits.dtype == torch.bfloat16 and gloss.numel() == 1
    _need(dlogits, R, ldd, ldd, "vocab_ce dlogits")
    calls.append(("vocab_cross_entropy_bwd", R, Cn))


def _soft_kl_fwd(logits, target, row_label, lse, tsum, rowloss, loss, count, R, Cn):
    assert logits.dtype == torch.float32 and target.dtype == torch.float32 and row_label.dtype == torch.int64 and row_label.numel() == R
    _need(logits, R, logits.stride(0), Cn, "soft_kl logits"); _need(target, R, target.stride(0), Cn, "soft_kl target")
    assert lse.numel() >= R and tsum.numel() >= R and rowloss.numel() >= R and loss.numel() == 1 and count.numel() == 1
    calls.append(("soft_target_kl_fwd", R, Cn))


def _soft_kl_bwd(logits, target, row_label, lse, tsum, count, gloss, dlogits, ldd, R, Cn):
    assert ldd % 8 == 0 and ldd >= Cn and dlogits.dtype == torch.bfloat16 and gloss.numel() == 1
    _need(dlogits, R, ldd, ldd, "soft_kl dlogits")
    calls.append(("soft_target_kl_bwd", R, Cn))


def _copy_rows(src, src_bstride, dst, dst_bstride, nb, rpb, H):
    assert H % 8 == 0
    _need(src, (nb - 1) * src_bstride + rpb, H, H, "copy_rows src"); _need(dst, (nb - 1) * dst_bstride + rpb, H, H, "copy_rows dst")


def _l2norm_fwd(x, ldx, y, ldy, inv, rows, D, eps=1e-12):
    assert x.dtype in (torch.float32, torch.bfloat16) and y.dtype == torch.bfloat16 and inv.dtype == torch.float32
    _need(x, rows, ldx, D, "l2norm x"); _need(y, rows, ldy, D, "l2norm y"); assert inv.numel() >= rows and inv.is_contiguous()


def _l2norm_bwd(g, ldg, y, ldy, inv, dx, lddx, rows, D):
    _need(g, rows, ldg, D, "l2norm g"); _need(y, rows, ldy, D, "l2norm y"); _need(dx, rows, lddx, D, "l2norm dx")


def _gather_rows2(a, b, idx, out, n, H):
    assert H % 8 == 0 and a.shape[1] == H and b.shape[1] == H and idx.dtype == torch.int64 and idx.numel() == n
    assert int(idx.min()) >= 0 and int(idx.max()) < a.shape[0] + b.shape[0]
    _need(out, n, H, H, "gather2 out")


def _ptr_fwd(q, k, mask_add, out, ldo, B, T, Nn, HQ, scale):
    assert HQ % 8 == 0
    _need(q, B * T, HQ, HQ, "ptr q"); _need(k, B * Nn, HQ, HQ, "ptr k"); _need(out, B * T, ldo, Nn, "ptr out")
    assert mask_add is None or (mask_add.dtype == torch.float32 and mask_add.numel() == B * Nn)


def _ptr_bwd(ds, ldd, q, k, dq, dk, B, T, Nn, HQ, scale):
    assert ds.dtype == torch.float32
    _need(ds, B * T, ldd, Nn, "ptr ds"); _need(dq, B * T, HQ, HQ, "ptr dq"); _need(dk, B * Nn, HQ, HQ, "ptr dk")


def _scatter_add(x, ld, nb, rpb, bstride, idx, idx_ld, per_pos, idx_base, out, H, few_buckets, skip_bucket=-1):
    _need(x, (nb - 1) * bstride + rpb, ld, H, "scatter_add x")
    if idx is not None:
        assert idx.dtype == torch.int64 and _room(idx) >= (nb - 1) * idx_ld + rpb
        assert int(idx.max()) < out.shape[0], "scatter_add: index %d outside the %d-row table" % (int(idx.max()), out.shape[0])
    assert out.dtype == torch.float32 and out.shape[1] == H


def _embed_tables_bwd(x, ld, B, T, R, seg, vt, pos0, dpos, dtyp, dtyp_vis, dpos_vis, H):
    calls.append(("embed_tables_bwd",))
    _need(x, B * (T + R), ld, H, "embed_tables_bwd x")
    assert seg is None or (seg.dtype == torch.int64 and seg.numel() == B * T)
    assert (R == 0) == (vt is None) or dtyp_vis is None
    assert vt is None or (vt.dtype == torch.int64 and vt.numel() == B * R)
    for t, need in ((dpos, pos0 + T), (dtyp, 1), (dtyp_vis, 1), (dpos_vis, 1)):
        assert t is None or (t.dtype == torch.float32 and t.shape[1] == H and t.shape[0] >= need and t.is_contiguous())
    if seg is not None and dtyp is not None:
        assert int(seg.max()) < dtyp.shape[0]
    if vt is not None and dtyp_vis is not None:
        assert int(vt.max()) < dtyp_vis.shape[0]


def _cast2d_f32(src, lds, dst, ldd, rows, cols):
    assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16
    _need(src, rows, lds, cols, "cast2d src"); _need(dst, rows, ldd, ldd, "cast2d dst")


def _bce_rowmask_fwd(scores, targets, w, loss, count, rows, Nn):
    assert scores.numel() == rows * Nn == targets.numel() and w.numel() == rows and loss.numel() == 1 and count.numel() == 1


def _bce_rowmask_bwd(scores, targets, w, count, gloss, d, rows, Nn):
    assert d.numel() == rows * Nn and gloss.numel() == 1


def _mse_fwd(pred, target, loss, rows, cols, row_label=None, count=None):
    assert pred.dtype == torch.float32 == target.dtype and loss.numel() == 1
    assert row_label is None or (row_label.dtype == torch.int64 and row_label.numel() == rows and count is not None and count.numel() == 1)
    _need(pred, rows, pred.stride(0), cols, "mse pred"); _need(target, rows, target.stride(0), cols, "mse target")


def _mse_bwd(pred, target, gloss, d, ldd, rows, cols, row_label=None, count=None):
    assert d.dtype == torch.bfloat16 and ldd % 8 == 0 and ldd >= cols and gloss.numel() == 1
    _need(d, rows, ldd, ldd, "mse dpred")


def _wra_common(seq, ld, B, S, H, M, Nn, txt_pad, img_pad, label, xinv, yinv, plan, cost, dist):
    assert seq.dtype == torch.bfloat16 and M <= 128 and Nn <= 128 and S >= M + Nn and ld >= H
    _need(seq, B * S, ld, H, "wra seq")
    assert txt_pad.numel() == B * M and img_pad.numel() == B * Nn and label.numel() == B and label.dtype == torch.int64
    assert xinv.numel() == B * M and yinv.numel() == B * Nn and plan.numel() == B * M * Nn == cost.numel() and dist.numel() == B


def _wra_fwd(seq, ld, B, S, H, M, Nn, txt_pad, img_pad, label, xinv, yinv, plan, cost, dist, loss, count):
    _wra_common(seq, ld, B, S, H, M, Nn, txt_pad, img_pad, label, xinv, yinv, plan, cost, dist)
    assert loss.numel() == 1 and count.numel() == 1


def _wra_bwd(seq, ld, B, S, H, M, Nn, txt_pad, img_pad, label, xinv, yinv, plan, cost, dist, gloss, count, dseq, ldd):
    _wra_common(seq, ld, B, S, H, M, Nn, txt_pad, img_pad, label, xinv, yinv, plan, cost, dist)
    assert dseq.dtype == torch.bfloat16 and gloss.numel() == 1
    _need(dseq, B * S, ldd, H, "wra dseq")


def _visual_masks(input_mask, image_dim, B, T, R, image_mask, attention_mask, vtype, mask_add, pool_index):
    """The one stub that has to PRODUCE values: the index tensors it writes are validated by the other checkers downstream."""
    assert input_mask.dtype == torch.int64 and tuple(input_mask.shape) == (B, T)
    assert image_dim is None or (image_dim.dtype == torch.int64 and image_dim.numel() == B)
    dim = torch.full((B, 1), R, dtype=torch.int64) if image_dim is None else image_dim.reshape(B, 1)
    image_mask.copy_((torch.arange(R).expand(B, R) < dim).long())
    attention_mask.copy_(torch.cat((input_mask, image_mask), dim=1))
    vtype.zero_()
    mask_add.copy_((1.0 - attention_mask.float()) * -10000.0)
    pool_index.copy_(input_mask.sum(1) - 2)
    calls.append(("visual_masks", B, T, R))


def _expand_batch(x, out, Bs, reps, n, mode):
    assert x.dtype in (torch.bfloat16, torch.float32) and out.dtype == x.dtype and mode in (0, 1)
    assert n % (8 if x.dtype == torch.bfloat16 else 4) == 0
    assert _room(x) >= Bs * n and _room(out) >= reps * Bs * n, "expand_batch: %d x %d x %d does not fit" % (reps, Bs, n)
    calls.append(("expand_batch" if x.dtype == torch.bfloat16 else "expand_batch_f32", Bs, reps, n, mode))


def _reduce_batch(g, dx, Bs, reps, n, mode):
    assert g.dtype in (torch.bfloat16, torch.float32) and dx.dtype == g.dtype and mode in (0, 1)
    assert _room(g) >= reps * Bs * n and _room(dx) >= Bs * n
    calls.append(("reduce_batch" if g.dtype == torch.bfloat16 else "reduce_batch_f32", Bs, reps, n, mode))


_CHECKED = {"expand_batch": _expand_batch, "reduce_batch": _reduce_batch, "visual_masks": _visual_masks, "mse_fwd": _mse_fwd, "mse_bwd": _mse_bwd, "wra_fwd": _wra_fwd, "wra_bwd": _wra_bwd, "soft_target_kl_fwd": _soft_kl_fwd, "soft_target_kl_bwd": _soft_kl_bwd, "vocab_cross_entropy_fwd": _vocab_ce_fwd, "vocab_cross_entropy_bwd": _vocab_ce_bwd, "gemm_f32": _gemm_f32, "attention_f32_fwd": _attention_f32_fwd, "attention_f32_bwd": _attention_f32_bwd, "layernorm_f32_fwd": _layernorm_f32_fwd,
            "embed_text_f32_fwd": _embed_text_f32, "gather_rows_f32": _gather_rows_f32, "gemm": _gemm, "gemm_grouped": _gemm_grouped, "attention_fwd": _attention_fwd, "attention_bwd": _attention_bwd, "copy_rows": _copy_rows,
            "l2norm_rows_fwd": _l2norm_fwd, "l2norm_rows_bwd": _l2norm_bwd, "gather_rows2": _gather_rows2, "ptr_scores_fwd": _ptr_fwd,
            "ptr_scores_bwd": _ptr_bwd, "rows_scatter_add": _scatter_add, "embed_tables_bwd": _embed_tables_bwd, "cast2d_f32_to_bf16": _cast2d_f32,
            "bce_rowmask_fwd": _bce_rowmask_fwd, "bce_rowmask_bwd": _bce_rowmask_bwd}


@contextlib.contextmanager
def installed():
    """Replace every kernel wrapper of mmf_amd._native by its checker (or a no-op) for the duration of the block."""
    saved = {}
    for name, obj in list(vars(N).items()):
        if name.startswith("__") or name in _KEEP or not callable(obj) or isinstance(obj, type):
            continue
        if getattr(obj, "__module__", None) != N.__name__:
            continue
        saved[name] = obj
        if name in _CHECKED:
            setattr(N, name, _CHECKED[name])
        elif name in _INT_RETURNS:
            setattr(N, name, _INT_RETURNS[name])
        else:
            setattr(N, name, (lambda nm: (lambda *a, **k: calls.append((nm,))))(name))
    del calls[:]
    from mmf_amd import functional as Fn
    try:
        # dropout keys normally come from the device's Philox state: use the graph-mode key sequence instead
        with Fn.dropout_keys.graph_mode(torch.zeros(1, dtype=torch.int32)):
            yield calls
    finally:
        Fn.shadows.clear()
        for name, obj in saved.items():
            setattr(N, name, obj)
