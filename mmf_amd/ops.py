"""PyTorch custom-op surface of the MI355X kernels: `torch.ops.mmf_amd.*`.

SURVEY.md §8(b): MMF has no FFI on this path — what a native replacement must export is a PyTorch custom-op
ABI (`torch.ops` operators over `Tensor`s on the current HIP stream, registered autograd, scriptable) behind the
`nn.Module`s that keep the reference's parameter names.  The operators are declared here with `torch.library`
(schemas below; TorchScript and the dispatcher see them like any built-in op), their implementation is the
autograd surface of mmf_amd/functional.py (every forward AND backward a hand-written gfx950 kernel behind the C ABI
of include/mmf_amd.h), and autograd is registered by construction: each op is a `CompositeImplicitAutograd` kernel
that applies the corresponding `torch.autograd.Function`, so gradients flow in eager mode, under `torch.jit.script`
(the reference's own tests script the model: tests/models/test_visual_bert.py:40-49, tests/test_utils.py:270-297)
and through `torch.jit.save` / `load`.

Each op takes the fp32 master parameters under the reference's names; the bf16 weight shadows (and the packed
Q|K|V views) are looked up inside (mmf_amd.functional.ShadowCache), dropout keys are drawn inside from `p` and
`training`, so a scripted module holds nothing but Tensors, ints, floats and bools.

    torch.ops.mmf_amd.transformer_layer   BertLayerJit.forward                         mmf/modules/hf_layers.py:255-292
    torch.ops.mmf_amd.visio_linguistic_embeddings   BertVisioLinguisticEmbeddings.forward      mmf/modules/embeddings.py:423-459
    torch.ops.mmf_amd.additive_mask       (1 - mask) * -10000                           mmf/models/visual_bert.py:94-106
    torch.ops.mmf_amd.gather_rows         `vqa` pooling gather + dropout               mmf/models/visual_bert.py:389-400
    torch.ops.mmf_amd.dense_gelu          HF BertIntermediate / head transform dense    (call sites hf_layers.py:289, visual_bert.py:328)
    torch.ops.mmf_amd.layer_norm          nn.LayerNorm                                  visual_bert.py:328
    torch.ops.mmf_amd.linear              nn.Linear                                     visual_bert.py:330
    torch.ops.mmf_amd.linear_tanh         HF BertPooler                                 visual_bert.py:146
    torch.ops.mmf_amd.dropout             nn.Dropout                                    visual_bert.py:400
    torch.ops.mmf_amd.pair_halves         nlvr2 pooled-output pairing                   visual_bert.py:369-374
    torch.ops.mmf_amd.masked_lm_head      tied decoder + masked-LM CrossEntropyLoss      visual_bert.py:267-277
    torch.ops.mmf_amd.masked_region_head  image-prediction decoder + masked KLDivLoss    vilbert.py:846-858,1150-1157
    torch.ops.mmf_amd.logit_bce           LogitBinaryCrossEntropy                       mmf/modules/losses.py:225-251

Inside `with mmf_amd.fp32_inference():` every operator above routes to the fp32-accurate forward kernels instead
(mmf_amd/fp32_path.py: fp32 activations, fp32-input MFMA; north_star's 1e-3 bound) — same schemas, same modules.
"""
from typing import Optional

import torch

from mmf_amd import _ops_native
from mmf_amd import fp32_path as F32P
from mmf_amd import fp32_train as F32T
from mmf_amd import functional as Fn

# Where the operators live: with the native library loaded (any GPU box) the schemas and the kernels of the operators in
# _ops_native.NATIVE_OPS come from libmmf_amd_ops.so (mmf_amd/csrc/torch_ops.cpp, C++ autograd nodes), and the functions below are bound
# as their `_py_` twins — entered only inside `mmf_amd.fp32_inference()` or with an experiment hook of mmf_amd/utils/graph.py active —
# or, for the pretraining heads, as the operator's kernel.  Without a GPU (dry runs against kernel stubs) everything is declared here.
NATIVE = _ops_native.NATIVE
LIB = torch.library.Library("mmf_amd", "IMPL" if NATIVE else "DEF")
_SCHEMAS = {}


def _op(schema):
    name = schema.split("(")[0]

    def deco(fn):
        if not NATIVE:
            LIB.define(schema)
            LIB.impl(name, fn, "CompositeImplicitAutograd")
        elif name in _ops_native.PY_TWINS:
            LIB.impl("_py_" + name, fn, "CompositeImplicitAutograd")
        elif name not in _ops_native.NATIVE_OPS:
            LIB.impl(name, fn, "CompositeImplicitAutograd")        # (schema defined by the native library, kernel bound here)
        _SCHEMAS[name] = schema
        return fn
    return deco


# (Operand checks — device, dtype, contiguity, extents — are made where the pointers are taken: mmf_amd/_native.py raises
# NativeLibraryError for a host tensor or a wrong dtype, the C ABI's MMF_CHECK_ARG for shapes; there is no CPU path.)


@_op("visual_masks(Tensor input_mask, Tensor? image_dim, int R) -> (Tensor, Tensor, Tensor, Tensor, Tensor)")
def visual_masks(input_mask, image_dim, R):
    """(image_mask, attention_mask, visual_embeddings_type, additive mask, `vqa` pooling index) of VisualBERT.forward's input massaging
    (visual_bert.py:444-467, 525-556, 389-392) in one launch."""
    im = input_mask.long().contiguous()
    B, T = im.shape
    dim = None if image_dim is None else image_dim.long().reshape(-1).contiguous()
    image_mask = torch.empty(B, R, dtype=torch.int64, device=im.device)
    attention_mask = torch.empty(B, T + R, dtype=torch.int64, device=im.device)
    vtype = torch.empty(B, R, dtype=torch.int64, device=im.device)
    mask_add = torch.empty(B, T + R, dtype=torch.float32, device=im.device)
    pool = torch.empty(B, dtype=torch.int64, device=im.device)
    Fn.nat.visual_masks(im, dim, B, T, R, image_mask, attention_mask, vtype, mask_add, pool)
    return image_mask, attention_mask, vtype, mask_add, pool


@_op("additive_mask(Tensor mask) -> Tensor")
def additive_mask(mask):
    am = mask.contiguous()
    if am.dtype != torch.int64:
        am = am.long()
    out = torch.empty(am.shape, dtype=torch.float32, device=am.device)
    Fn.nat.make_additive_mask(am, out)
    return out


@_op("visio_linguistic_embeddings(Tensor input_ids, Tensor token_type_ids, Tensor? visual_embeddings, Tensor? visual_embeddings_type, "
     "Tensor word, Tensor pos, Tensor typ, Tensor ln_w, Tensor ln_b, Tensor typ_vis, Tensor pos_vis, Tensor proj_w, Tensor proj_b, "
     "float eps, float p, bool training, int pad_idx, Tensor? image_text_alignment=None) -> Tensor")
def visio_linguistic_embeddings(input_ids, token_type_ids, visual_embeddings, visual_embeddings_type, word, pos, typ, ln_w, ln_b,
                                typ_vis, pos_vis, proj_w, proj_b, eps, p, training, pad_idx, image_text_alignment=None):
    if visual_embeddings is None or visual_embeddings_type is None:
        visual_embeddings = visual_embeddings_type = image_text_alignment = None
    if F32T.active():
        return F32T.visio_linguistic_embeddings(input_ids, token_type_ids, visual_embeddings, visual_embeddings_type, word, pos, typ, ln_w, ln_b,
                                                typ_vis, pos_vis, proj_w, proj_b, eps, p, training, pad_idx, image_text_alignment)
    if F32P.active():
        F32P.check_no_dropout(p, training)
        return F32P.visio_linguistic_embeddings(input_ids, token_type_ids, visual_embeddings, visual_embeddings_type, word, pos, typ,
                                                ln_w, ln_b, typ_vis, pos_vis, proj_w, proj_b, eps, image_text_alignment)
    w16 = Fn.shadows.get(proj_w) if visual_embeddings is not None else None
    return Fn.VisioLinguisticEmbeddingsFn.apply(
        input_ids, token_type_ids, visual_embeddings, visual_embeddings_type, word, pos, typ, ln_w, ln_b, typ_vis, pos_vis, proj_w,
        proj_b, w16, eps, Fn.make_drop(p, training), pad_idx if pad_idx >= 0 else None, image_text_alignment)


@_op("transformer_layer(Tensor x, Tensor wq, Tensor bq, Tensor wk, Tensor bk, Tensor wv, Tensor bv, Tensor wo, Tensor bo, "
     "Tensor ln1_w, Tensor ln1_b, Tensor w1, Tensor b1, Tensor w2, Tensor b2, Tensor ln2_w, Tensor ln2_b, Tensor? mask_add, "
     "int heads, float eps1, float eps2, float p_attn, float p_hid1, float p_hid2, bool training, int causal_tail) -> Tensor")
def transformer_layer(x, wq, bq, wk, bk, wv, bv, wo, bo, ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b, mask_add, heads, eps1, eps2,
                      p_attn, p_hid1, p_hid2, training, causal_tail):
    if F32T.active():
        return F32T.transformer_layer(x, wq, bq, wk, bk, wv, bv, wo, bo, ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b, mask_add, heads, eps1, eps2,
                                      p_attn, p_hid1, p_hid2, training, causal_tail)
    if F32P.active():
        for p in (p_attn, p_hid1, p_hid2):
            F32P.check_no_dropout(p, training)
        return F32P.transformer_layer(x, wq, bq, wk, bk, wv, bv, wo, bo, ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b, mask_add, heads,
                                      eps1, eps2, causal_tail)
    wqkv16 = Fn.shadows.get(wq, wk, wv)
    bqkv = Fn.shadows.get(bq, bk, bv, dtype=torch.float32)
    mask = mask_add
    if mask is not None and causal_tail > 0:
        mask = Fn.PrefixLMMask(mask, causal_tail)
    return Fn.TransformerLayerFn.apply(
        x, wq, bq, wk, bk, wv, bv, wo, bo, ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b, wqkv16, bqkv, Fn.shadows.get(wo),
        Fn.shadows.get(w1), Fn.shadows.get(w2), mask, heads, eps1, eps2, Fn.make_drop(p_attn, training), Fn.make_drop(p_hid1, training),
        Fn.make_drop(p_hid2, training))


@_op("linear(Tensor x, Tensor weight, Tensor? bias, bool out_f32) -> Tensor")
def linear(x, weight, bias, out_f32):
    if F32T.active():
        return F32T.linear(x, weight, bias)
    if F32P.active():
        return F32P.linear(x, weight, bias)
    return Fn.LinearFn.apply(x, weight, bias, Fn.shadows.get(weight), out_f32)


@_op("layer_norm(Tensor x, Tensor weight, Tensor bias, float eps) -> Tensor")
def layer_norm(x, weight, bias, eps):
    if F32T.active():
        return F32T.layer_norm(x, weight, bias, eps)
    if F32P.active():
        return F32P.layer_norm(x, weight, bias, eps)
    return Fn.LayerNormFn.apply(x, weight, bias, eps)


@_op("dense_gelu(Tensor x, Tensor weight, Tensor bias) -> Tensor")
def dense_gelu(x, weight, bias):
    if F32T.active():
        return F32T.dense_gelu(x, weight, bias)
    if F32P.active():
        return F32P.dense_gelu(x, weight, bias)
    return Fn.DenseGeluFn.apply(x, weight, bias, Fn.shadows.get(weight))


@_op("linear_tanh(Tensor x, Tensor weight, Tensor bias) -> Tensor")
def linear_tanh(x, weight, bias):
    if F32T.active():
        return F32T.linear_tanh(x, weight, bias)
    if F32P.active():
        return F32P.linear_tanh(x, weight, bias)
    return Fn.LinearTanhFn.apply(x, weight, bias, Fn.shadows.get(weight))


@_op("gather_rows(Tensor x, Tensor index, float p, bool training) -> Tensor")
def gather_rows(x, index, p, training):
    if F32T.active():
        return F32T.gather_rows(x, index, p, training)
    if F32P.active():
        F32P.check_no_dropout(p, training)
        return F32P.gather_rows(x, index)
    return Fn.GatherRowsFn.apply(x, index, Fn.make_drop(p, training))


@_op("dropout(Tensor x, float p, bool training) -> Tensor")
def dropout(x, p, training):
    if F32T.active():
        return F32T.dropout(x, p, training)
    if F32P.active():
        F32P.check_no_dropout(p, training)
        return x
    drop = Fn.make_drop(p, training)
    if not drop[1]:
        return x
    return Fn.DropoutFn.apply(x, drop)


@_op("pair_halves(Tensor x) -> Tensor")
def pair_halves(x):
    if F32T.active():
        return F32T.pair_halves(x)
    if F32P.active():
        return F32P.pair_halves(x)
    return Fn.PairHalvesFn.apply(x)


@_op("masked_lm_head(Tensor x, Tensor weight, Tensor bias, Tensor labels, int ignore_index) -> (Tensor, Tensor)")
def masked_lm_head(x, weight, bias, labels, ignore_index):
    if F32T.active():
        return F32T.masked_lm_head(x, weight, bias, labels, ignore_index)
    if F32P.active():
        return F32P.masked_lm_head(x, weight, bias, labels, ignore_index)
    return Fn.MaskedLMHeadFn.apply(x, weight, bias, Fn.shadows.get(weight), labels, ignore_index)


@_op("masked_region_head(Tensor x, Tensor weight, Tensor bias, Tensor target, Tensor row_label) -> (Tensor, Tensor)")
def masked_region_head(x, weight, bias, target, row_label):
    if F32T.active():
        return F32T.masked_region_head(x, weight, bias, target, row_label)
    if F32P.active():
        return F32P.masked_region_head(x, weight, bias, target, row_label)
    return Fn.MaskedRegionHeadFn.apply(x, weight, bias, Fn.shadows.get(weight), target, row_label)


@_op("logit_bce(Tensor scores, Tensor targets) -> Tensor")
def logit_bce(scores, targets):
    if F32T.active():
        return F32T.logit_bce(scores, targets)
    return Fn.LogitBCEFn.apply(scores, targets)


def schemas():
    """name -> schema string of every registered operator (tests)."""
    return dict(_SCHEMAS)
