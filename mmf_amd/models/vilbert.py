"""ViLBERT behind MMF's model API on the gfx950 kernels (BASELINE.json configs[2]).

Mirrors mmf/models/vilbert.py: `BertImageFeatureEmbeddings` (:891-913), `BertBiAttention` (:347-475), `BertBiOutput`
(:478-512), `BertConnectionLayer` (:515-556), `BertEncoder` (:559-796), `BertTextPooler` / `BertImagePooler` (:799-826),
`ViLBERTBase` (:916-1051), `ViLBERTForClassification` (:1243-1333) and the registered `ViLBERT(BaseModel)` (:1336-1472):
same config keys, same `forward(sample_list) -> {"scores": [B, num_labels]}`, and the reference's parameter tree
(`model.bert.embeddings.*`, `.v_embeddings.*`, `.encoder.{layer,v_layer,c_layer}.*`, `.t_pooler`, `.v_pooler`,
`model.classifier.*`, including the never-used `biOutput.q_dense{1,2}`), so MMF checkpoints load unmodified.

MI355X-first internals: the text / visual stream layers are the same two fused autograd nodes as VisualBERT's layers
(the visual stream and the co-attention run the head_dim-128 build of the attention kernel); a connection layer is one
bi-attention node (two packed Q|K|V GEMMs + two cross attentions that read the other stream's K, V in place), two
dense+dropout+residual+LayerNorm nodes and two feed-forward nodes.

The `nlvr2` head (two images per sample, :1262-1265, 1322-1323, 1369-1394) is built.
The pretraining head (:1054-1240, `visual_target` 0 / 1 / 2), `fixed_{t,v}_layer` (:625-666) and the `in_batch_pairs` / `fast_mode` batch
expansion (:678-725, `expand_batch_kernel` / `reduce_batch_kernel`; bf16 path) are built too.
Not built (raise): `task_specific_tokens`, `in_batch_pairs` / `fast_mode` together with `dynamic_attention`, and asking the inner modules for the
attention maps (`output_all_attention_masks=True`: the fused kernels never materialise them; the `visualization` config flag itself is accepted — through the
registered model the reference never returns the maps either).
"""
import os

import torch
from torch import nn

from mmf_amd import functional as Fn
from mmf_amd import fp32_path as F32P
from mmf_amd import fp32_train as F32T
from mmf_amd.common.registry import registry
from mmf_amd.models.base_model import BaseModel
from mmf_amd.modules.hf_layers import (
    BertConfig, BertEmbeddingsJit, BertIntermediate, BertLayerJit, BertLMPredictionHead, BertOutput, BertPredictionHeadTransform, LayerNorm,
    Linear,
    init_bert_weights)
from mmf_amd.utils.configuration import to_container
from mmf_amd.utils.modeling import get_optimizer_parameters_for_bert


def _stream_config(config, prefix):
    """BertConfig view of one stream: '' = text (hidden_size, ...), 'v_' = visual (v_hidden_size, ...).  Only the visual
    self-attention carries the `dynamic_attention` gates (vilbert.py:161,174-176), fed from the text stream's width."""
    g = lambda k, d=None: getattr(config, prefix + k, getattr(config, k, d))
    return BertConfig(hidden_size=g("hidden_size"), num_attention_heads=g("num_attention_heads"),
                      intermediate_size=g("intermediate_size"), hidden_dropout_prob=g("hidden_dropout_prob"),
                      attention_probs_dropout_prob=g("attention_probs_dropout_prob"), hidden_act=g("hidden_act", "gelu"),
                      layer_norm_eps=1e-12 if prefix else getattr(config, "layer_norm_eps", 1e-12),
                      dynamic_attention=bool(prefix) and bool(getattr(config, "dynamic_attention", False)),
                      dynamic_attention_input_size=config.hidden_size)


def _additive_mask(mask):
    m = mask.contiguous().long()
    out = torch.empty(m.shape, dtype=torch.float32, device=m.device)
    Fn.nat.make_additive_mask(m, out)      # (1 - m) * -10000, vilbert.py:1000,1009
    return out


class BertImageFeatureEmbeddings(nn.Module):
    """vilbert.py:891-913."""

    def __init__(self, config):
        super().__init__()
        self.image_embeddings = Linear(config.v_feature_size, config.v_hidden_size)
        self.image_location_embeddings = Linear(5, config.v_hidden_size)
        self.LayerNorm = LayerNorm(config.v_hidden_size, eps=1e-12)
        self.dropout_prob = config.hidden_dropout_prob

    def forward(self, image_feature, image_location):
        ie, le = self.image_embeddings, self.image_location_embeddings
        if F32T.active():      # mmf_amd.fp32_training(): fp32 forward + backward
            return F32T.image_feature_embeddings(image_feature, image_location, ie.weight, ie.bias, le.weight, le.bias, self.LayerNorm.weight,
                                                 self.LayerNorm.bias, self.LayerNorm.eps, self.dropout_prob, self.training)
        if F32P.active():      # fp32-accurate forward (mmf_amd.fp32_inference()): same operations on the fp32 kernels
            F32P.check_no_dropout(self.dropout_prob, self.training)
            return F32P.image_feature_embeddings(image_feature, image_location, ie.weight, ie.bias, le.weight, le.bias,
                                                 self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)
        return Fn.ImageFeatureEmbeddingsFn.apply(image_feature, image_location, ie.weight, ie.bias, le.weight, le.bias,
                                                 self.LayerNorm.weight, self.LayerNorm.bias, Fn.shadows.get(ie.weight),
                                                 self.LayerNorm.eps, Fn.make_drop(self.dropout_prob, self.training))


class BertBiAttention(nn.Module):
    """vilbert.py:347-475."""

    def __init__(self, config):
        super().__init__()
        if config.bi_hidden_size % config.bi_num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)" % (
                config.bi_hidden_size, config.bi_num_attention_heads))
        self.num_attention_heads = config.bi_num_attention_heads
        self.attention_head_size = config.bi_hidden_size // config.bi_num_attention_heads
        if self.attention_head_size not in (64, 128):
            raise ValueError("the gfx950 fused attention kernel is built for head_dim 64 or 128, got %d" % self.attention_head_size)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query1 = Linear(config.v_hidden_size, self.all_head_size)
        self.key1 = Linear(config.v_hidden_size, self.all_head_size)
        self.value1 = Linear(config.v_hidden_size, self.all_head_size)
        self.dropout1_prob = config.v_attention_probs_dropout_prob
        self.query2 = Linear(config.hidden_size, self.all_head_size)
        self.key2 = Linear(config.hidden_size, self.all_head_size)
        self.value2 = Linear(config.hidden_size, self.all_head_size)
        self.dropout2_prob = config.attention_probs_dropout_prob

    def forward(self, input_tensor1, attention_mask1, input_tensor2, attention_mask2, co_attention_mask=None,
                use_co_attention_mask=False):
        """(image, image additive mask [B, R], text, text additive mask [B, T]) -> (context_layer1 [B, T, bi],
        context_layer2 [B, R, bi], {})."""
        if use_co_attention_mask:
            raise NotImplementedError("use_co_attention_mask is dead code in the reference (vilbert.py:421,448) and is not built")
        if F32T.active():
            c1, c2 = F32T.bi_attention(input_tensor1, input_tensor2, self.query1, self.key1, self.value1, self.query2, self.key2, self.value2,
                                       attention_mask1, attention_mask2, self.num_attention_heads, self.dropout1_prob, self.dropout2_prob,
                                       self.training)
            return c1, c2, {}
        if F32P.active():
            F32P.check_no_dropout(max(self.dropout1_prob, self.dropout2_prob), self.training)
            c1, c2 = F32P.bi_attention(input_tensor1, input_tensor2, self.query1, self.key1, self.value1, self.query2, self.key2,
                                       self.value2, attention_mask1, attention_mask2, self.num_attention_heads)
            return c1, c2, {}
        s = Fn.shadows
        w1 = s.get(self.query1.weight, self.key1.weight, self.value1.weight)
        b1 = s.get(self.query1.bias, self.key1.bias, self.value1.bias, dtype=torch.float32)
        w2 = s.get(self.query2.weight, self.key2.weight, self.value2.weight)
        b2 = s.get(self.query2.bias, self.key2.bias, self.value2.bias, dtype=torch.float32)
        c1, c2 = Fn.BiAttentionFn.apply(
            input_tensor1, input_tensor2, self.query1.weight, self.query1.bias, self.key1.weight, self.key1.bias,
            self.value1.weight, self.value1.bias, self.query2.weight, self.query2.bias, self.key2.weight, self.key2.bias,
            self.value2.weight, self.value2.bias, w1, b1, w2, b2, attention_mask1, attention_mask2, self.num_attention_heads,
            Fn.make_drop(self.dropout1_prob, self.training), Fn.make_drop(self.dropout2_prob, self.training))
        return c1, c2, {}


class BertBiOutput(nn.Module):
    """vilbert.py:478-512 (q_dense1 / q_dense2 exist in the reference's parameter tree and are never called)."""

    def __init__(self, config):
        super().__init__()
        self.dense1 = Linear(config.bi_hidden_size, config.v_hidden_size)
        self.LayerNorm1 = LayerNorm(config.v_hidden_size, eps=1e-12)
        self.dropout1_prob = config.v_hidden_dropout_prob
        self.q_dense1 = Linear(config.bi_hidden_size, config.v_hidden_size)
        self.dense2 = Linear(config.bi_hidden_size, config.hidden_size)
        self.LayerNorm2 = LayerNorm(config.hidden_size, eps=1e-12)
        self.dropout2_prob = config.hidden_dropout_prob
        self.q_dense2 = Linear(config.bi_hidden_size, config.hidden_size)

    def forward(self, hidden_states1, input_tensor1, hidden_states2, input_tensor2):
        if F32T.active():
            return (F32T.dense_residual_ln(hidden_states1, input_tensor1, self.dense1.weight, self.dense1.bias, self.LayerNorm1.weight,
                                           self.LayerNorm1.bias, self.LayerNorm1.eps, self.dropout1_prob, self.training),
                    F32T.dense_residual_ln(hidden_states2, input_tensor2, self.dense2.weight, self.dense2.bias, self.LayerNorm2.weight,
                                           self.LayerNorm2.bias, self.LayerNorm2.eps, self.dropout2_prob, self.training))
        if F32P.active():
            F32P.check_no_dropout(max(self.dropout1_prob, self.dropout2_prob), self.training)
            return (F32P.dense_residual_ln(hidden_states1, input_tensor1, self.dense1.weight, self.dense1.bias, self.LayerNorm1.weight,
                                           self.LayerNorm1.bias, self.LayerNorm1.eps),
                    F32P.dense_residual_ln(hidden_states2, input_tensor2, self.dense2.weight, self.dense2.bias, self.LayerNorm2.weight,
                                           self.LayerNorm2.bias, self.LayerNorm2.eps))
        out1 = Fn.DenseDropoutResidualLNFn.apply(
            hidden_states1, input_tensor1, self.dense1.weight, self.dense1.bias, self.LayerNorm1.weight, self.LayerNorm1.bias,
            Fn.shadows.get(self.dense1.weight), self.LayerNorm1.eps, Fn.make_drop(self.dropout1_prob, self.training))
        out2 = Fn.DenseDropoutResidualLNFn.apply(
            hidden_states2, input_tensor2, self.dense2.weight, self.dense2.bias, self.LayerNorm2.weight, self.LayerNorm2.bias,
            Fn.shadows.get(self.dense2.weight), self.LayerNorm2.eps, Fn.make_drop(self.dropout2_prob, self.training))
        return out1, out2


def _feed_forward(it, ot, x, training):
    if F32T.active():
        return F32T.feed_forward(x, it.dense.weight, it.dense.bias, ot.dense.weight, ot.dense.bias, ot.LayerNorm.weight, ot.LayerNorm.bias,
                                 ot.LayerNorm.eps, ot.dropout_prob, training)
    if F32P.active():
        F32P.check_no_dropout(ot.dropout_prob, training)
        return F32P.feed_forward(x, it.dense.weight, it.dense.bias, ot.dense.weight, ot.dense.bias, ot.LayerNorm.weight, ot.LayerNorm.bias,
                                 ot.LayerNorm.eps)
    return Fn.FeedForwardFn.apply(x, it.dense.weight, it.dense.bias, ot.dense.weight, ot.dense.bias, ot.LayerNorm.weight,
                                  ot.LayerNorm.bias, Fn.shadows.get(it.dense.weight), Fn.shadows.get(ot.dense.weight),
                                  ot.LayerNorm.eps, Fn.make_drop(ot.dropout_prob, training))


class BertImageLayer(BertLayerJit):
    """vilbert.py:313-345: a visual-stream layer.  Without `dynamic_attention` it IS the shared encoder layer (one fused autograd
    node); with it the queries and keys are gated from the text stream (BertImageSelfAttention, vilbert.py:199-212): attention
    block with `qk_gate`, then the feed-forward block."""

    def forward(self, hidden_states, attention_mask=None, txt_embedding=None, txt_attention_mask=None):
        sa, so = self.attention.self, self.attention.output
        if not sa.dynamic_attention:
            return super().forward(hidden_states, attention_mask)
        B, S, _ = hidden_states.shape
        gate = sa.dynamic_gate(txt_embedding, txt_attention_mask)
        if F32T.active():      # mmf_amd.fp32_training(): the gate, the gated attention block and the feed-forward block on the fp32 kernels, with backward
            attention_output = F32T.attention_block(
                hidden_states, sa.query.weight, sa.query.bias, sa.key.weight, sa.key.bias, sa.value.weight, sa.value.bias, so.dense.weight,
                so.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias, attention_mask.reshape(B, S).float(), sa.num_attention_heads,
                so.LayerNorm.eps, sa.dropout_prob, so.dropout_prob, self.training, qk_gate=gate)
            return (_feed_forward(self.intermediate, self.output, attention_output, self.training),)
        if F32P.active():
            F32P.check_no_dropout(max(sa.dropout_prob, so.dropout_prob), self.training)
            attention_output = F32P.attention_block(
                hidden_states, sa.query.weight, sa.query.bias, sa.key.weight, sa.key.bias, sa.value.weight, sa.value.bias, so.dense.weight,
                so.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias, attention_mask.reshape(B, S).float(), sa.num_attention_heads,
                so.LayerNorm.eps, gate)
            return (_feed_forward(self.intermediate, self.output, attention_output, self.training),)
        w16, b32 = sa.packed_qkv()
        attention_output = Fn.AttentionBlockFn.apply(
            hidden_states, sa.query.weight, sa.query.bias, sa.key.weight, sa.key.bias, sa.value.weight, sa.value.bias,
            so.dense.weight, so.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias, w16, b32, Fn.shadows.get(so.dense.weight),
            attention_mask.reshape(B, S).float().contiguous(), sa.num_attention_heads, so.LayerNorm.eps,
            Fn.make_drop(sa.dropout_prob, self.training), Fn.make_drop(so.dropout_prob, self.training), gate)
        return (_feed_forward(self.intermediate, self.output, attention_output, self.training),)


# Two HIP streams for the two modality streams (round 3).  Between connection points the text layers and the visual layers are independent,
# and inside a connection layer so are the two output blocks and the two feed-forward blocks — and each of them alone under-fills the
# chip (B = 32: 4096 text rows x 768 and 3232 region rows x 1024 give 128 and 104 wide GEMM tiles for 256 CUs).  The visual side runs
# on a side stream forked from / joined to the caller's stream (captured as parallel branches by a hipGraph; autograd replays each
# node's backward on its forward stream, so the backward overlaps the same way).  MMF_AMD_VILBERT_STREAMS=0 keeps one stream (A/B).
_TWO_STREAMS = os.environ.get("MMF_AMD_VILBERT_STREAMS", "1") != "0"
_side_streams = {}


def _fork(x):
    """(main, side) streams when the visual side may run beside the text side, else None."""
    if not (_TWO_STREAMS and x.is_cuda) or F32P.active() or F32T.active():
        return None
    dev = x.device
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    return main, side


def _join(streams, *tensors):
    main, side = streams
    main.wait_stream(side)
    for t in tensors:
        t.record_stream(main)      # allocated on the side stream, consumed on the caller's from here on


class BertConnectionLayer(nn.Module):
    """vilbert.py:515-556."""

    def __init__(self, config):
        super().__init__()
        vcfg, tcfg = _stream_config(config, "v_"), _stream_config(config, "")
        self.biattention = BertBiAttention(config)
        self.biOutput = BertBiOutput(config)
        self.v_intermediate = BertIntermediate(vcfg)
        self.v_output = BertOutput(vcfg)
        self.t_intermediate = BertIntermediate(tcfg)
        self.t_output = BertOutput(tcfg)

    def forward(self, input_tensor1, attention_mask1, input_tensor2, attention_mask2, co_attention_mask=None,
                use_co_attention_mask=False):
        bi_output1, bi_output2, co_attention_probs = self.biattention(
            input_tensor1, attention_mask1, input_tensor2, attention_mask2, co_attention_mask, use_co_attention_mask)
        streams = _fork(input_tensor1)
        if streams is None:
            attention_output1, attention_output2 = self.biOutput(bi_output2, input_tensor1, bi_output1, input_tensor2)
            layer_output1 = _feed_forward(self.v_intermediate, self.v_output, attention_output1, self.training)
            layer_output2 = _feed_forward(self.t_intermediate, self.t_output, attention_output2, self.training)
            return layer_output1, layer_output2, co_attention_probs
        bo = self.biOutput
        with torch.cuda.stream(streams[1]):     # the visual half: output block + feed-forward block
            a1 = Fn.DenseDropoutResidualLNFn.apply(
                bi_output2, input_tensor1, bo.dense1.weight, bo.dense1.bias, bo.LayerNorm1.weight, bo.LayerNorm1.bias,
                Fn.shadows.get(bo.dense1.weight), bo.LayerNorm1.eps, Fn.make_drop(bo.dropout1_prob, self.training))
            layer_output1 = _feed_forward(self.v_intermediate, self.v_output, a1, self.training)
        a2 = Fn.DenseDropoutResidualLNFn.apply(
            bi_output1, input_tensor2, bo.dense2.weight, bo.dense2.bias, bo.LayerNorm2.weight, bo.LayerNorm2.bias,
            Fn.shadows.get(bo.dense2.weight), bo.LayerNorm2.eps, Fn.make_drop(bo.dropout2_prob, self.training))
        layer_output2 = _feed_forward(self.t_intermediate, self.t_output, a2, self.training)
        _join(streams, layer_output1)
        return layer_output1, layer_output2, co_attention_probs


class BertEncoder(nn.Module):
    """vilbert.py:559-796."""

    def __init__(self, config):
        super().__init__()
        # `in_batch_pairs` / `fast_mode` (vilbert.py:678-725): at the first connection point the batch becomes every text against every image
        self.in_batch_pairs = bool(getattr(config, "in_batch_pairs", False))
        self.fast_mode = bool(getattr(config, "fast_mode", False))
        if (self.in_batch_pairs or self.fast_mode) and bool(getattr(config, "dynamic_attention", False)):
            raise NotImplementedError("in_batch_pairs / fast_mode with dynamic_attention: the reference does not expand the mask its gates pool "
                                      "with (vilbert.py:678-725 leave extended_attention_mask2 at the old batch size)")
        # vilbert.py:625-666, as the reference BEHAVES: its loop runs `forward_no_grad` on the layer at t_start and sets t_start = fixed_t_layer
        # in that same iteration, so only the first such layer executes (without gradient) and layers t_start + 1 .. fixed_t_layer - 1 never run
        self.fixed_t_layer = int(getattr(config, "fixed_t_layer", 0) or 0)
        self.fixed_v_layer = int(getattr(config, "fixed_v_layer", 0) or 0)
        self.with_coattention = config.with_coattention
        self.v_biattention_id = list(config.v_biattention_id)
        self.t_biattention_id = list(config.t_biattention_id)
        self.layer = nn.ModuleList([BertLayerJit(_stream_config(config, "")) for _ in range(config.num_hidden_layers)])
        self.v_layer = nn.ModuleList([BertImageLayer(_stream_config(config, "v_")) for _ in range(config.v_num_hidden_layers)])
        self.c_layer = nn.ModuleList([BertConnectionLayer(config) for _ in range(len(self.v_biattention_id))])

    def forward(self, txt_embedding, image_embedding, txt_attention_mask, txt_attention_mask2, image_attention_mask,
                co_attention_mask=None, output_all_encoded_layers=True, output_all_attention_masks=False):
        """Masks are the additive fp32 key masks [B, T] / [B, R]; `txt_attention_mask2` is the 0 / 1 text mask [B, T] the
        dynamic_attention gates pool the text stream with (the reference's extended_attention_mask2, vilbert.py:985)."""
        if output_all_attention_masks:
            raise NotImplementedError("attention maps are never materialised by the fused kernel")
        v_start = t_start = 0
        all_t, all_v = [], []
        dynamic = any(l.attention.self.dynamic_attention for l in self.v_layer)     # then a visual layer reads the text stream: no overlap
        for count, (v_end, t_end) in enumerate(zip(self.v_biattention_id, self.t_biattention_id)):
            assert self.fixed_t_layer <= t_end and self.fixed_v_layer <= v_end            # vilbert.py:622-623
            if t_start < self.fixed_t_layer:
                with torch.no_grad():
                    txt_embedding = self.layer[t_start](txt_embedding, txt_attention_mask)[0]
                t_start = self.fixed_t_layer
            if v_start < self.fixed_v_layer:
                # (reference order: after this block's text layers; with dynamic_attention the gate reads the text stream as it is then)
                if dynamic:
                    for idx in range(t_start, t_end):
                        txt_embedding = self.layer[idx](txt_embedding, txt_attention_mask)[0]
                    t_start = t_end
                with torch.no_grad():
                    image_embedding = self.v_layer[v_start](image_embedding, image_attention_mask, txt_embedding, txt_attention_mask2)[0]
                v_start = self.fixed_v_layer
            streams = _fork(image_embedding) if (not dynamic and v_end > v_start and t_end > t_start) else None
            if streams is not None:
                with torch.cuda.stream(streams[1]):
                    for idx in range(v_start, v_end):
                        image_embedding = self.v_layer[idx](image_embedding, image_attention_mask, txt_embedding, txt_attention_mask2)[0]
                for idx in range(t_start, t_end):
                    txt_embedding = self.layer[idx](txt_embedding, txt_attention_mask)[0]
                _join(streams, image_embedding)
            else:
                for idx in range(t_start, t_end):
                    txt_embedding = self.layer[idx](txt_embedding, txt_attention_mask)[0]
                for idx in range(v_start, v_end):
                    image_embedding = self.v_layer[idx](image_embedding, image_attention_mask, txt_embedding, txt_attention_mask2)[0]
            if count == 0 and (self.in_batch_pairs or self.fast_mode):
                # (the same broadcast on whichever arithmetic is active: bf16 activations, or the fp32 kernels of mmf_amd.fp32_training() / fp32_inference())
                expand = F32T.expand_batch if F32T.active() else (F32P.expand_batch if F32P.active() else Fn.ExpandBatchFn.apply)
                if self.in_batch_pairs:          # new batch size = batch_size ^ 2 (:678-710): pair (i, j) = text i with image j
                    B = txt_embedding.shape[0]
                    image_embedding = expand(image_embedding, B, 0)
                    image_attention_mask = image_attention_mask.unsqueeze(0).expand(B, B, -1).reshape(B * B, -1).contiguous()
                    txt_embedding = expand(txt_embedding, B, 1)
                    txt_attention_mask = txt_attention_mask.unsqueeze(1).expand(B, B, -1).reshape(B * B, -1).contiguous()
                if self.fast_mode:               # one text against N images (:712-723)
                    N = image_embedding.shape[0]
                    if txt_embedding.shape[0] != N:
                        if txt_embedding.shape[0] != 1:
                            raise ValueError("fast_mode expands a text batch of 1 to the image batch (got %d texts, %d images)" % (txt_embedding.shape[0], N))
                        txt_embedding = expand(txt_embedding, N, 0)
                        txt_attention_mask = txt_attention_mask.expand(N, -1).contiguous()
            if self.with_coattention:
                image_embedding, txt_embedding, _ = self.c_layer[count](image_embedding, image_attention_mask, txt_embedding,
                                                                        txt_attention_mask)
            v_start, t_start = v_end, t_end
            if output_all_encoded_layers:
                all_t.append(txt_embedding)
                all_v.append(image_embedding)
        streams = _fork(image_embedding) if (not dynamic and v_start < len(self.v_layer) and t_start < len(self.layer)) else None
        if streams is not None:
            with torch.cuda.stream(streams[1]):
                for idx in range(v_start, len(self.v_layer)):
                    image_embedding = self.v_layer[idx](image_embedding, image_attention_mask, txt_embedding, txt_attention_mask2)[0]
            for idx in range(t_start, len(self.layer)):
                txt_embedding = self.layer[idx](txt_embedding, txt_attention_mask)[0]
            _join(streams, image_embedding)
        else:
            for idx in range(v_start, len(self.v_layer)):
                image_embedding = self.v_layer[idx](image_embedding, image_attention_mask, txt_embedding, txt_attention_mask2)[0]
            for idx in range(t_start, len(self.layer)):
                txt_embedding = self.layer[idx](txt_embedding, txt_attention_mask)[0]
        if not output_all_encoded_layers:
            all_t.append(txt_embedding)
            all_v.append(image_embedding)
        return all_t, all_v, ([], [], [])


class _ReluPooler(nn.Module):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.dense = Linear(in_dim, out_dim)

    def forward(self, hidden_states):
        B = hidden_states.shape[0]
        index = torch.zeros(B, dtype=torch.int64, device=hidden_states.device)
        if F32T.active():
            return F32T.relu(self.dense(F32T.GatherRowsFn.apply(hidden_states, index)))
        if F32P.active():
            return F32P.relu(self.dense(F32P.gather_rows(hidden_states, index)))
        first = Fn.GatherRowsFn.apply(hidden_states, index, Fn.nat.NO_DROP)
        return Fn.ReluFn.apply(self.dense(first))


class BertTextPooler(_ReluPooler):
    """vilbert.py:799-811."""

    def __init__(self, config):
        super().__init__(config.hidden_size, config.bi_hidden_size)


class BertImagePooler(_ReluPooler):
    """vilbert.py:814-826."""

    def __init__(self, config):
        super().__init__(config.v_hidden_size, config.bi_hidden_size)


class ViLBERTBase(nn.Module):
    """vilbert.py:916-1051."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        if getattr(config, "task_specific_tokens", False):
            # the reference's own path fails here: it extends the text mask by one token (vilbert.py:971-974) that its embeddings never add (no caller passes
            # task_ids, and HF's BertEmbeddings would take them as position_ids, :1018) -> a size mismatch in the first text layer
            raise NotImplementedError("task_specific_tokens (vilbert.py:971-974): the reference path itself fails with a size mismatch; not built")
        # `visualization: true` is accepted: in the reference it only makes the attention modules return their probabilities in `attn_data`
        # (vilbert.py:105-114, 238-247, 465-475), which the encoder collects when `output_all_attention_masks` is set (:599-795) — an argument the registered
        # model never passes (ViLBERT.forward -> self.model(...), :1445-1455), so through MMF's interface the flag changes no output.  Asking the inner
        # modules for the maps directly (`output_all_attention_masks=True`) raises in ViLBERTBase / BertEncoder: the fused kernels do not materialise them.
        self.embeddings = BertEmbeddingsJit(config)
        self.v_embeddings = BertImageFeatureEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.t_pooler = BertTextPooler(config)
        self.v_pooler = BertImagePooler(config)
        self.init_weights()

    def _init_weights(self, module):
        init_bert_weights(module, self.config.initializer_range)

    def init_weights(self):
        self.apply(self._init_weights)

    def forward(self, input_txt, image_feature, image_location, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, co_attention_mask=None, task_ids=None, output_all_encoded_layers=False,
                output_all_attention_masks=False):
        if attention_mask is None:
            attention_mask = torch.ones_like(input_txt)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_txt)
        if image_attention_mask is None:
            image_attention_mask = torch.ones(image_feature.size(0), image_feature.size(1)).type_as(input_txt)
        txt_mask_add = _additive_mask(attention_mask)
        img_mask_add = _additive_mask(image_attention_mask)
        embedding_output = self.embeddings(input_txt, token_type_ids)
        v_embedding_output = self.v_embeddings(image_feature, image_location)
        encoded_layers_t, encoded_layers_v, all_attention_mask = self.encoder(
            embedding_output, v_embedding_output, txt_mask_add, attention_mask.float(), img_mask_add, None,
            output_all_encoded_layers=output_all_encoded_layers, output_all_attention_masks=output_all_attention_masks)
        sequence_output_t = encoded_layers_t[-1]
        sequence_output_v = encoded_layers_v[-1]
        pooled_output_t = self.t_pooler(sequence_output_t)
        pooled_output_v = self.v_pooler(sequence_output_v)
        return (sequence_output_t, sequence_output_v, pooled_output_t, pooled_output_v, None,
                encoded_layers_t if output_all_encoded_layers else None, encoded_layers_v if output_all_encoded_layers else None)


class BertImagePredictionHead(nn.Module):
    """vilbert.py:829-858: BertImgPredictionHeadTransform (dense -> gelu -> LayerNorm(eps=1e-12) at the visual width) + decoder onto
    the `v_target_size` region classes."""

    def __init__(self, config):
        super().__init__()
        if config.get("hidden_act", "gelu") != "gelu":
            raise NotImplementedError("BertImgPredictionHeadTransform: only hidden_act='gelu' is fused in the GEMM epilogue")
        self.transform = BertPredictionHeadTransform(BertConfig(hidden_size=config.v_hidden_size, layer_norm_eps=1e-12))
        self.decoder = Linear(config.v_hidden_size, config.v_target_size)


class ViLBERTPreTrainingHeads(nn.Module):
    """vilbert.py:861-888 `BertPreTrainingHeads`: masked-LM head (decoder tied to the word embeddings), masked-region head and the
    image-text matching classifier `bi_seq_relationship` — whose score the reference computes and never uses (the
    next-sentence loss is commented out, :1236-1239): its parameters exist for checkpoints and receive no gradient."""

    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)
        self.bi_seq_relationship = Linear(config.bi_hidden_size, 2)
        self.imagePredictions = BertImagePredictionHead(config)
        self.fusion_method = config.fusion_method


class ViLBERTForPretraining(nn.Module):
    """vilbert.py:1054-1240: masked language modelling on the text stream + masked region classification on the visual stream."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.bert_config = BertConfig.from_dict({k: v for k, v in to_container(config).items()})
        for k in ("v_biattention_id", "t_biattention_id"):
            setattr(self.bert_config, k, list(config[k]))
        self.bert = ViLBERTBase(self.bert_config)
        self.cls = ViLBERTPreTrainingHeads(self.bert_config)
        self.vocab_size = self.config.vocab_size
        self.visual_target = config.visual_target
        self.num_negative = config.num_negative
        if self.visual_target not in (0, 1, 2):
            raise NotImplementedError("visual_target=%r: 0 (KL masked-region classification, the reference default), 1 (masked-region regression) "
                                      "and 2 (NCE with sampled negatives) are the forms of vilbert.py:1070-1075" % (self.visual_target,))
        self.init_weights()

    def init_weights(self):
        if self.config.get("random_initialize", False) is False:
            self.cls.apply(self.bert._init_weights)
            self.tie_weights()

    def tie_weights(self):
        """vilbert.py:1088-1095."""
        self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight

    def negative_index(self, batch_size, num_regions, like):
        """The sampled negatives of `visual_target: 2` (vilbert.py:1158-1203): flat region indices [B, R, num_negative], 70 % drawn from the
        OTHER samples of the batch, 30 % from the other regions of the same image — the same sequence of `random_` draws on tensors of the
        same shapes as the reference (index massaging on the host side of the boundary, as there)."""
        n_across, n_inside = int(self.num_negative * 0.7), int(self.num_negative * 0.3)
        B, R = batch_size, num_regions
        assert B != 0
        row_across = torch.ones(B, R, n_across, dtype=like.dtype, device=like.device).random_(0, B - 1)
        col_across = torch.ones(B, R, n_across, dtype=like.dtype, device=like.device).random_(0, R)
        own = torch.arange(B, device=like.device, dtype=like.dtype).view(B, 1, 1)
        row_across = torch.where(row_across == own, torch.full_like(row_across, B - 1), row_across)      # (row B - 1 never draws itself: values < B - 1)
        col_inside = torch.ones(B, R, n_inside, dtype=like.dtype, device=like.device).random_(0, R - 1)
        region = torch.arange(R, device=like.device, dtype=like.dtype).view(1, R, 1)
        col_inside = torch.where((col_inside == region) & (region < R - 1), torch.full_like(col_inside, R - 1), col_inside)
        return torch.cat((row_across * R + col_across, own * R + col_inside), dim=2)

    def forward(self, input_ids, image_feature, image_location, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, masked_lm_labels=None, image_label=None, image_target=None,
                output_all_attention_masks=False):
        (sequence_output_t, sequence_output_v, _, _, _, _, _) = self.bert(
            input_ids, image_feature, image_location, token_type_ids, attention_mask, image_attention_mask,
            output_all_encoded_layers=False, output_all_attention_masks=output_all_attention_masks)
        output = {}
        if image_label is not None and image_target is not None:
            head = self.cls.imagePredictions
            hidden_v = head.transform(sequence_output_v)
            if self.visual_target == 2:          # NCE against sampled negatives, CrossEntropyLoss on class 0, :1158-1227
                B, R = image_label.shape[0], image_label.shape[1]
                neg = self.negative_index(B, R, input_ids)
                if F32T.active():      # mmf_amd.fp32_training(): the same draws, decoder and loss on the fp32 kernels
                    img_loss, _ = F32T.masked_region_nce(hidden_v, head.decoder.weight, head.decoder.bias, image_target, image_label, neg)
                else:
                    img_loss, _ = Fn.MaskedRegionNCEFn.apply(hidden_v, head.decoder.weight, head.decoder.bias, Fn.shadows.get(head.decoder.weight),
                                                         image_target, image_label, neg)
            elif self.visual_target == 1 and F32T.active():
                img_loss, _ = F32T.masked_region_regression(hidden_v, head.decoder.weight, head.decoder.bias, image_target, image_label)
            elif self.visual_target == 1:        # nn.MSELoss(reduction="none") over the masked regions / max(their element count, 1), :1139-1148
                img_loss, _ = Fn.MaskedRegionRegressionFn.apply(hidden_v, head.decoder.weight, head.decoder.bias, Fn.shadows.get(head.decoder.weight),
                                                                image_target, image_label)
            else:
                img_loss, _ = torch.ops.mmf_amd.masked_region_head(hidden_v, head.decoder.weight, head.decoder.bias, image_target, image_label)
            output["masked_img_loss"] = img_loss.unsqueeze(0)
        if masked_lm_labels is not None:
            heads = self.cls.predictions
            hidden_t = heads.transform(sequence_output_t)
            lm_loss, _ = torch.ops.mmf_amd.masked_lm_head(hidden_t, heads.decoder.weight, heads.bias, masked_lm_labels, -1)
            output["masked_lm_loss"] = lm_loss.unsqueeze(0)
        return output


class ViLBERTForClassification(nn.Module):
    """vilbert.py:1243-1333."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.bert_config = BertConfig.from_dict({k: v for k, v in to_container(config).items()})
        for k in ("v_biattention_id", "t_biattention_id"):
            setattr(self.bert_config, k, list(config[k]))
        self.bert = ViLBERTBase(self.bert_config)
        self.training_head_type = self.config.training_head_type
        self.num_labels = self.config.num_labels
        self.fusion_method = config.fusion_method
        if self.fusion_method not in ("sum", "mul"):
            raise AssertionError
        self.dropout_prob = self.config.hidden_dropout_prob
        head_width = config.bi_hidden_size * (2 if self.training_head_type == "nlvr2" else 1)      # vilbert.py:1262-1265
        ccfg = BertConfig(hidden_size=head_width, layer_norm_eps=self.bert_config.layer_norm_eps)
        self.classifier = nn.Sequential(BertPredictionHeadTransform(ccfg), Linear(head_width, self.num_labels))
        self.init_weights()

    def init_weights(self):
        if self.config.get("random_initialize", False) is False:
            self.classifier.apply(self.bert._init_weights)

    def forward(self, input_ids, image_feature, image_location, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, masked_lm_labels=None, image_label=None, image_target=None,
                next_sentence_label=None, output_all_attention_masks=False):
        (sequence_output_t, sequence_output_v, pooled_output_t, pooled_output_v, _, _, _) = self.bert(
            input_ids, image_feature, image_location, token_type_ids, attention_mask, image_attention_mask,
            output_all_encoded_layers=False, output_all_attention_masks=output_all_attention_masks)
        output = {}
        if F32T.active():
            fused = (F32T.eltwise_mul if self.fusion_method == "mul" else F32T.add)(pooled_output_t, pooled_output_v)
            fused = F32T.dropout(fused, self.dropout_prob, self.training)
        elif F32P.active():
            F32P.check_no_dropout(self.dropout_prob, self.training)
            fused = (F32P.eltwise_mul if self.fusion_method == "mul" else F32P.add)(pooled_output_t, pooled_output_v)
        elif self.fusion_method == "mul":
            fused = Fn.EltwiseMulFn.apply(pooled_output_t, pooled_output_v)
        else:
            fused = pooled_output_t + pooled_output_v
        drop = Fn.nat.NO_DROP if (F32T.active() or F32P.active()) else Fn.make_drop(self.dropout_prob, self.training)
        pooled_output = Fn.DropoutFn.apply(fused, drop) if drop[1] else fused
        if self.training_head_type == "nlvr2":
            # pairs CONSECUTIVE rows of the stacked [img0 batch; img1 batch] exactly as the reference's view does (:1322-1323)
            pooled_output = pooled_output.reshape(-1, pooled_output.size(1) * 2)
        hidden = self.classifier[0](pooled_output)
        logits = self.classifier[1](hidden, out_f32=True)
        output["scores"] = logits.contiguous().view(-1, self.num_labels)
        return output


@registry.register_model("vilbert")
class ViLBERT(BaseModel):
    """vilbert.py:1336-1472."""

    def __init__(self, config):
        super().__init__(config)

    @classmethod
    def config_path(cls):
        return "configs/models/vilbert/pretrain.yaml"

    @classmethod
    def format_state_key(cls, key):
        return (key.replace("bert.bert", "model.bert").replace("bert.cls", "model.cls")
                .replace("bert.classifier", "model.classifier"))

    def build(self):
        if self.config.training_head_type == "pretraining":
            self.model = ViLBERTForPretraining(self.config)
        else:
            self.model = ViLBERTForClassification(self.config)
        if self.config.get("freeze_base", False):
            for p in self.model.bert.parameters():
                p.requires_grad = False

    def get_image_and_text_features(self, sample_list):
        """vilbert.py:1364-1418 (single-image datasets)."""
        ids, mask, tt = sample_list["input_ids"], sample_list["input_mask"], sample_list["segment_ids"]
        if sample_list.get("dataset_name", None) == "nlvr2":          # vilbert.py:1369-1394
            ids, mask, tt = torch.cat([ids, ids]), torch.cat([mask, mask]), torch.cat([tt, tt])
            i0, i1 = sample_list["img0"], sample_list["img1"]
            info0, info1 = i0.get("image_info_0", None) or {}, i1.get("image_info_0", None) or {}
            return {
                "input_ids": ids, "attention_mask": mask, "token_type_ids": tt,
                "image_dim": torch.cat([info0["max_features"], info1["max_features"]]),
                "image_feature": torch.cat([i0["image_feature_0"], i1["image_feature_0"]]),
                "image_location": torch.cat([info0["bbox"], info1["bbox"]]), "image_target": None, "image_label": None,
            }
        image_info = sample_list.get("image_info_0", None) or {}
        return {
            "input_ids": ids, "attention_mask": mask, "token_type_ids": tt, "image_dim": image_info.get("max_features", None),
            "image_feature": sample_list.get("image_feature_0", None), "image_location": image_info.get("bbox", None),
            "image_target": self._image_target(image_info.get("cls_prob", None), ids.device),
            "image_label": sample_list.get("image_labels", None),
        }

    @staticmethod
    def _image_target(cls_prob, device):
        """vilbert.py:1402-1406: the detector's class distribution per region, as a float tensor on the model's device."""
        if cls_prob is None:
            return None
        if isinstance(cls_prob, torch.Tensor):
            return cls_prob.to(device=device, dtype=torch.float32)
        import numpy as np
        return torch.tensor(np.array(cls_prob, dtype=np.float32), dtype=torch.float, device=device)

    def get_optimizer_parameters(self, config):
        return get_optimizer_parameters_for_bert(self.model, config)

    def forward(self, sample_list):
        params = self.get_image_and_text_features(sample_list)
        params["masked_lm_labels"] = sample_list.get("lm_label_ids", None)
        if params["image_feature"] is not None and params["image_dim"] is not None:
            image_mask = torch.arange(params["image_feature"].size(-2), device=params["image_feature"].device).expand(
                *params["image_feature"].size()[:-1])
            if len(params["image_dim"].size()) < len(image_mask.size()):
                params["image_dim"] = params["image_dim"].unsqueeze(-1)
                assert len(params["image_dim"].size()) == len(image_mask.size())
            params["image_attention_mask"] = (image_mask < params["image_dim"]).long()
        else:
            params["image_attention_mask"] = None
        params.pop("image_dim")
        output_dict = self.model(params["input_ids"], params["image_feature"], params["image_location"], params["token_type_ids"],
                                 params["attention_mask"], params["image_attention_mask"], params["masked_lm_labels"],
                                 params["image_label"], params["image_target"])
        if self.config.training_head_type == "pretraining":           # vilbert.py:1459-1469
            loss_key = "{}/{}".format(sample_list["dataset_name"], sample_list["dataset_type"])
            output_dict["losses"] = {
                loss_key + "/masked_lm_loss": output_dict.pop("masked_lm_loss"),
                loss_key + "/masked_img_loss": output_dict.pop("masked_img_loss"),
            }
        return output_dict
