"""HIP-backed transformer encoder modules with the reference's class names, parameter names and
shapes (mmf/modules/hf_layers.py:138-355 plus the HF blocks it instantiates: BertSelfOutput,
BertIntermediate, BertOutput, BertPooler, BertPredictionHeadTransform).

`state_dict()` keys are identical to the reference's (`attention.self.query.weight`,
`attention.output.LayerNorm.bias`, `intermediate.dense.weight`, ...), so MMF / HF checkpoints load
unmodified.  Activations are bf16 `[B, S, H]` tensors in HBM; every forward and backward is a
hand-written gfx950 kernel (mmf_amd/functional.py).  There is no eager fallback.
"""
from typing import List, Optional, Tuple

import torch
from torch import Tensor, nn

from mmf_amd import functional as Fn
from mmf_amd import fp32_path as F32P
from mmf_amd import fp32_train as F32T
from mmf_amd import ops  # noqa: F401  (registers torch.ops.mmf_amd.*)


class BertConfig:
    """The subset of HF `BertConfig` the encoder reads (defaults = bert-base-uncased)."""

    def __init__(self, **kw):
        d = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                 hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=512,
                 type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-12, pad_token_id=0, output_attentions=False,
                 output_hidden_states=False, is_decoder=False)
        d.update(kw)
        for k, v in d.items():
            setattr(self, k, v)

    @classmethod
    def from_dict(cls, d):
        return cls(**{k: v for k, v in dict(d).items() if not isinstance(v, (dict, list))})

    def get(self, k, default=None):
        return getattr(self, k, default)


class Linear(nn.Module):
    """nn.Linear parameter container (weight [out, in], bias [out]) whose forward is the MFMA GEMM."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None

    def forward(self, x: Tensor, out_f32: bool = False) -> Tensor:
        return torch.ops.mmf_amd.linear(x, self.weight, self.bias, out_f32)


class Dropout(nn.Module):
    """nn.Dropout on a bf16 activation with the counter-hash mask of the fused kernels."""

    def __init__(self, p=0.1):
        super().__init__()
        self.p = p

    def forward(self, x: Tensor) -> Tensor:
        return torch.ops.mmf_amd.dropout(x, self.p, self.training)


class LayerNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.eps = eps

    def forward(self, x: Tensor) -> Tensor:
        return torch.ops.mmf_amd.layer_norm(x, self.weight, self.bias, self.eps)


def init_bert_weights(module, std=0.02):
    """HF `BertPreTrainedModel._init_weights`: normal(0, initializer_range) for Linear / Embedding
    weights, zero biases, LayerNorm = (1, 0)."""
    if isinstance(module, (Linear, nn.Linear)):
        module.weight.data.normal_(mean=0.0, std=std)
        if module.bias is not None:
            module.bias.data.zero_()
    elif isinstance(module, nn.Embedding):
        module.weight.data.normal_(mean=0.0, std=std)
        if module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()
    elif isinstance(module, (LayerNorm, nn.LayerNorm)):
        module.weight.data.fill_(1.0)
        module.bias.data.zero_()


def additive_key_mask(attention_mask, batch, seq):
    """The reference hands the encoder the broadcastable additive mask `[B, 1, 1, S]`
    (visual_bert.py:94-106).  The fused kernel wants it as fp32 `[B, S]`."""
    if attention_mask is None:
        return None
    if isinstance(attention_mask, Fn.PrefixLMMask):   # M4C's prefix-LM mask (m4c.py:424-440): key mask + causal tail
        return Fn.PrefixLMMask(additive_key_mask(attention_mask.key_mask, batch, seq), attention_mask.causal_tail)
    m = attention_mask
    if m.dim() == 4:
        if m.shape[1] != 1:         # one [S, S] mask per head (mmf_attn_desc.mask_head_stride): handed on as it is, [B, heads, S, S]
            if m.shape[0] != batch or tuple(m.shape[2:]) != (seq, seq):
                raise ValueError("attention_mask of shape %s does not match hidden states [%d, %d, ...]" % (tuple(m.shape), batch, seq))
            return m.float().contiguous()
        if m.shape[2] != 1:
            # A materialised additive mask per (query, key) pair, as `attention_scores + attention_mask` takes it (hf_layers.py:187-190;
            # MMT.forward builds one, m4c.py:424-440): the kernels read it from global memory (mmf_attn_desc.mask_query_stride).  For M4C's
            # prefix-LM structure `mmf_amd.functional.PrefixLMMask(key_mask, dec_steps)` is the cheaper form (nothing is materialised).
            if tuple(m.shape) != (batch, 1, seq, seq):
                raise ValueError("attention_mask of shape %s does not match hidden states [%d, %d, ...]" % (tuple(m.shape), batch, seq))
            m = m.reshape(batch, seq, seq)
        else:
            m = m.reshape(batch, seq)
    if m.dtype != torch.float32:
        m = m.float()
    return m.contiguous()


class BertSelfAttentionJit(nn.Module):
    """hf_layers.py:138-213.  Returns `(context_layer, attention_probs)`; the probabilities are never
    materialised by the fused kernel, so the second element is an empty placeholder."""

    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)" % (
                config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        if self.attention_head_size not in (64, 128):
            raise ValueError("the gfx950 fused attention kernel is built for head_dim 64 or 128, got %d" % self.attention_head_size)
        self.query = Linear(config.hidden_size, self.all_head_size)
        self.key = Linear(config.hidden_size, self.all_head_size)
        self.value = Linear(config.hidden_size, self.all_head_size)
        self.dropout_prob = config.attention_probs_dropout_prob
        # ViLBERT's BertImageSelfAttention with `dynamic_attention` (mmf/models/vilbert.py:174-176): gates from the text stream
        self.dynamic_attention = bool(getattr(config, "dynamic_attention", False))
        if self.dynamic_attention:
            self.dyLinear_q = Linear(config.dynamic_attention_input_size, self.all_head_size)
            self.dyLinear_k = Linear(config.dynamic_attention_input_size, self.all_head_size)

    @torch.jit.unused
    def dynamic_gate(self, txt_embedding, txt_attention_mask):
        """fp32 [B, 2 * all_head_size]: 1 + sigmoid(dyLinear_{q,k}(masked mean of the text stream)) (vilbert.py:204-209)."""
        if F32P.active():      # fp32-accurate forward (mmf_amd.fp32_inference())
            pool = F32P.masked_mean(txt_embedding, txt_attention_mask)
            return F32P.dynamic_gate(F32P.linear(pool, self.dyLinear_q.weight, self.dyLinear_q.bias),
                                     F32P.linear(pool, self.dyLinear_k.weight, self.dyLinear_k.bias))
        pool = (F32T.masked_mean if F32T.active() else Fn.MaskedMeanFn.apply)(txt_embedding, txt_attention_mask)   # (fp32 training: fp32 rows both ways)
        zq = torch.ops.mmf_amd.linear(pool, self.dyLinear_q.weight, self.dyLinear_q.bias, True)
        zk = torch.ops.mmf_amd.linear(pool, self.dyLinear_k.weight, self.dyLinear_k.bias, True)
        return Fn.DynamicGateFn.apply(zq, zk)

    def packed_qkv(self):
        w16 = Fn.shadows.get(self.query.weight, self.key.weight, self.value.weight)
        b32 = Fn.shadows.get(self.query.bias, self.key.bias, self.value.bias, dtype=torch.float32)
        return w16, b32

    @torch.jit.unused      # (a scripted BertLayerJit calls torch.ops.mmf_amd.transformer_layer; this sub-module forward stays eager-only)
    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None):
        if head_mask is not None or encoder_hidden_states is not None:
            raise NotImplementedError("head_mask / cross-attention are not on the VisualBERT path")
        B, S, _ = hidden_states.shape
        w16, b32 = self.packed_qkv()
        drop = Fn.make_drop(self.dropout_prob, self.training)
        ctx = Fn.SelfAttentionFn.apply(hidden_states, self.query.weight, self.query.bias, self.key.weight, self.key.bias,
                                       self.value.weight, self.value.bias, w16, b32,
                                       additive_key_mask(attention_mask, B, S), self.num_attention_heads, drop)
        return ctx, hidden_states.new_empty(0)


class BertSelfOutput(nn.Module):
    """HF BertSelfOutput: LayerNorm(dropout(dense(h)) + input)."""

    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout_prob = config.hidden_dropout_prob

    @torch.jit.unused      # (a scripted BertLayerJit calls torch.ops.mmf_amd.transformer_layer; this sub-module forward stays eager-only)
    def forward(self, hidden_states, input_tensor):
        drop = Fn.make_drop(self.dropout_prob, self.training)
        return Fn.DenseDropoutResidualLNFn.apply(hidden_states, input_tensor, self.dense.weight, self.dense.bias,
                                                 self.LayerNorm.weight, self.LayerNorm.bias, Fn.shadows.get(self.dense.weight),
                                                 self.LayerNorm.eps, drop)


class BertIntermediate(nn.Module):
    """HF BertIntermediate: gelu(dense(x)) with the exact-erf GELU."""

    def __init__(self, config):
        super().__init__()
        if getattr(config, "hidden_act", "gelu") != "gelu":
            raise ValueError("only hidden_act == 'gelu' is implemented")
        self.dense = Linear(config.hidden_size, config.intermediate_size)

    @torch.jit.unused      # (a scripted BertLayerJit calls torch.ops.mmf_amd.transformer_layer; this sub-module forward stays eager-only)
    def forward(self, hidden_states):
        return Fn.DenseGeluFn.apply(hidden_states, self.dense.weight, self.dense.bias, Fn.shadows.get(self.dense.weight))


class BertOutput(nn.Module):
    """HF BertOutput: LayerNorm(dropout(dense(h)) + input)."""

    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout_prob = config.hidden_dropout_prob

    @torch.jit.unused      # (a scripted BertLayerJit calls torch.ops.mmf_amd.transformer_layer; this sub-module forward stays eager-only)
    def forward(self, hidden_states, input_tensor):
        drop = Fn.make_drop(self.dropout_prob, self.training)
        return Fn.DenseDropoutResidualLNFn.apply(hidden_states, input_tensor, self.dense.weight, self.dense.bias,
                                                 self.LayerNorm.weight, self.LayerNorm.bias, Fn.shadows.get(self.dense.weight),
                                                 self.LayerNorm.eps, drop)


class BertAttentionJit(nn.Module):
    """hf_layers.py:216-252."""

    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttentionJit(config)
        self.output = BertSelfOutput(config)

    @torch.jit.unused      # (a scripted BertLayerJit calls torch.ops.mmf_amd.transformer_layer; this sub-module forward stays eager-only)
    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None):
        if head_mask is not None or encoder_hidden_states is not None:
            raise NotImplementedError("head_mask / cross-attention are not on the VisualBERT path")
        B, S, _ = hidden_states.shape
        sa, so = self.self, self.output
        w16, b32 = sa.packed_qkv()
        out = Fn.AttentionBlockFn.apply(
            hidden_states, sa.query.weight, sa.query.bias, sa.key.weight, sa.key.bias, sa.value.weight, sa.value.bias,
            so.dense.weight, so.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias, w16, b32, Fn.shadows.get(so.dense.weight),
            additive_key_mask(attention_mask, B, S), sa.num_attention_heads, so.LayerNorm.eps,
            Fn.make_drop(sa.dropout_prob, self.training), Fn.make_drop(so.dropout_prob, self.training), None)
        return (out,)


class BertLayerJit(nn.Module):
    """hf_layers.py:255-292: attention sub-layer then feed-forward sub-layer as ONE autograd node (Fn.TransformerLayerFn),
    so that backward can run the layer's four weight gradients as one grouped launch."""

    def __init__(self, config):
        super().__init__()
        self.attention = BertAttentionJit(config)
        self.is_decoder = bool(getattr(config, "is_decoder", False))
        if self.is_decoder:      # hf_layers.py:268-271: built, never called by forward (:273-292) — parameters a decoder-mode checkpoint carries
            self.crossattention = BertAttentionJit(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states: Tensor, attention_mask: Optional[Tensor] = None, head_mask: Optional[Tensor] = None,
                encoder_hidden_states: Optional[Tensor] = None, encoder_attention_mask: Optional[Tensor] = None) -> Tuple[Tensor]:
        """One `torch.ops.mmf_amd.transformer_layer` call.  `attention_mask`: the additive mask the reference hands the encoder,
        `[B, 1, 1, S]` (visual_bert.py:94-106) or already `[B, S]`; eager callers may pass M4C's prefix-LM mask as a
        `functional.PrefixLMMask` (key mask + number of causally visible decoding steps, m4c.py:424-440)."""
        if head_mask is not None or encoder_hidden_states is not None:
            raise NotImplementedError("head_mask / cross-attention are not on the VisualBERT path")
        B, S = hidden_states.shape[0], hidden_states.shape[1]
        causal_tail = 0
        if not torch.jit.is_scripting():
            if isinstance(attention_mask, Fn.PrefixLMMask):
                causal_tail = attention_mask.causal_tail
                attention_mask = attention_mask.key_mask
        mask_add: Optional[Tensor] = None
        if attention_mask is not None:
            m = attention_mask
            if m.dim() == 4:
                if m.shape[1] != 1:       # one [S, S] mask per head, [B, heads, S, S] (mmf_attn_desc.mask_head_stride)
                    m = m.reshape(B, -1, S, S)
                elif m.shape[2] != 1:     # a materialised additive mask per (query, key) pair (see additive_key_mask)
                    m = m.reshape(B, S, S)
                else:
                    m = m.reshape(B, S)
            mask_add = m.float().contiguous()
        sa, so = self.attention.self, self.attention.output
        it, ot = self.intermediate, self.output
        layer_output = torch.ops.mmf_amd.transformer_layer(
            hidden_states, sa.query.weight, sa.query.bias, sa.key.weight, sa.key.bias, sa.value.weight, sa.value.bias,
            so.dense.weight, so.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias,
            it.dense.weight, it.dense.bias, ot.dense.weight, ot.dense.bias, ot.LayerNorm.weight, ot.LayerNorm.bias,
            mask_add, sa.num_attention_heads, so.LayerNorm.eps, ot.LayerNorm.eps,
            sa.dropout_prob, so.dropout_prob, ot.dropout_prob, self.training, causal_tail)
        return (layer_output,)


class BertEncoderJit(nn.Module):
    """hf_layers.py:295-355."""

    def __init__(self, config):
        super().__init__()
        self.output_attentions = getattr(config, "output_attentions", False)
        self.output_hidden_states = getattr(config, "output_hidden_states", False)
        self.layer = nn.ModuleList([BertLayerJit(config) for _ in range(config.num_hidden_layers)])

    def forward(self, hidden_states: Tensor, attention_mask: Optional[Tensor] = None, encoder_hidden_states: Optional[Tensor] = None,
                encoder_attention_mask: Optional[Tensor] = None, output_attentions: bool = False,
                output_hidden_states: bool = False, return_dict: bool = False, head_mask: Optional[Tensor] = None) -> Tuple[Tensor]:
        """Typed like the reference's scriptable encoder (hf_layers.py:317-355): `Tuple[Tensor]` under TorchScript; the eager
        call appends the tuple of all hidden states when `output_hidden_states` is set."""
        if output_attentions:
            raise NotImplementedError("attention probabilities are not materialised by the fused kernel")
        all_hidden_states = ()
        for layer_module in self.layer:
            if not torch.jit.is_scripting() and output_hidden_states:
                all_hidden_states = all_hidden_states + (hidden_states,)
            hidden_states = layer_module(hidden_states, attention_mask, None, encoder_hidden_states, encoder_attention_mask)[0]
        if not torch.jit.is_scripting() and output_hidden_states:
            all_hidden_states = all_hidden_states + (hidden_states,)
        outputs = (hidden_states,)
        if not torch.jit.is_scripting():
            if output_hidden_states:
                outputs = outputs + (all_hidden_states,)
        return outputs


class BertPooler(nn.Module):
    """HF BertPooler: tanh(dense(h[:, 0])).  Dead work under `pooler_strategy: vqa` (visual_bert.py:146
    vs :389-398); kept for the `default` strategy and for checkpoint compatibility."""

    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.hidden_size, config.hidden_size)

    def forward(self, hidden_states: Tensor) -> Tensor:
        # row 0 of every sample (gather kernel), then dense + tanh in one GEMM epilogue
        B = hidden_states.shape[0]
        index = torch.zeros(B, dtype=torch.int64, device=hidden_states.device)
        first = torch.ops.mmf_amd.gather_rows(hidden_states, index, 0.0, False)
        return torch.ops.mmf_amd.linear_tanh(first, self.dense.weight, self.dense.bias)


class BertEmbeddingsJit(nn.Module):
    """HF BertEmbeddings parameters (hf_layers.py:98-135).  `forward` is the text-only embedding stage."""

    def __init__(self, config):
        super().__init__()
        H = config.hidden_size
        self.word_embeddings = nn.Embedding(config.vocab_size, H, padding_idx=getattr(config, "pad_token_id", 0))
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, H)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, H)
        self.LayerNorm = LayerNorm(H, eps=config.layer_norm_eps)
        self.dropout_prob = config.hidden_dropout_prob

    def forward(self, input_ids, token_type_ids=None, position_ids=None, inputs_embeds=None):
        if position_ids is not None or inputs_embeds is not None:
            raise NotImplementedError("explicit position_ids / inputs_embeds are not on the built paths")
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        if F32T.active():      # mmf_amd.fp32_training()
            return F32T.visio_linguistic_embeddings(input_ids, token_type_ids, None, None, self.word_embeddings.weight, self.position_embeddings.weight,
                                                    self.token_type_embeddings.weight, self.LayerNorm.weight, self.LayerNorm.bias, None, None, None, None,
                                                    self.LayerNorm.eps, self.dropout_prob, self.training, self.word_embeddings.padding_idx)
        if F32P.active():      # fp32-accurate forward (mmf_amd.fp32_inference())
            F32P.check_no_dropout(self.dropout_prob, self.training)
            return F32P.visio_linguistic_embeddings(input_ids, token_type_ids, None, None, self.word_embeddings.weight,
                                                    self.position_embeddings.weight, self.token_type_embeddings.weight, self.LayerNorm.weight,
                                                    self.LayerNorm.bias, None, None, None, None, self.LayerNorm.eps)
        z = self.word_embeddings.weight.new_zeros(1, self.word_embeddings.weight.shape[1])
        return Fn.VisioLinguisticEmbeddingsFn.apply(
            input_ids, token_type_ids, None, None, self.word_embeddings.weight, self.position_embeddings.weight,
            self.token_type_embeddings.weight, self.LayerNorm.weight, self.LayerNorm.bias, z, z, z, z, None,
            self.LayerNorm.eps, Fn.make_drop(self.dropout_prob, self.training), self.word_embeddings.padding_idx)


class BertModelJit(nn.Module):
    """Parameter tree of the reference's BertModelJit (hf_layers.py:358-452): embeddings / encoder / pooler."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = BertEmbeddingsJit(config)
        self.encoder = BertEncoderJit(config)
        self.pooler = BertPooler(config)
        self.apply(lambda m: init_bert_weights(m, config.initializer_range))

    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    def forward(self, input_ids, attention_mask=None, token_type_ids=None):
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        am = attention_mask.contiguous().long()
        mask_add = torch.empty(am.shape, dtype=torch.float32, device=am.device)
        Fn.nat.make_additive_mask(am, mask_add)
        h = self.embeddings(input_ids, token_type_ids)
        seq = self.encoder(h, mask_add.view(am.shape[0], 1, 1, am.shape[1]))[0]
        return seq, self.pooler(seq)


class BertPredictionHeadTransform(nn.Module):
    """HF BertPredictionHeadTransform: LayerNorm(gelu(dense(x))); `in_dim` as in MMF's PredictionHeadTransformWithInDim
    (mmf/models/transformers/heads/mlp.py:90-94)."""

    def __init__(self, config, in_dim=None):
        super().__init__()
        self.dense = Linear(config.hidden_size if in_dim is None else in_dim, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, hidden_states: Tensor) -> Tensor:
        h = torch.ops.mmf_amd.dense_gelu(hidden_states, self.dense.weight, self.dense.bias)
        return self.LayerNorm(h)


class BertLMPredictionHead(nn.Module):
    """HF BertLMPredictionHead as pinned by the reference (transformers <= 4.10.1): transform, a bias-free decoder whose
    `bias` attribute IS `self.bias` (one tensor under the two state-dict keys `predictions.bias` and `predictions.decoder.bias`)."""

    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.decoder.bias = self.bias


class BertPreTrainingHeads(nn.Module):
    """HF BertPreTrainingHeads: masked-LM head + the next-sentence classifier (whose score the reference computes and never
    uses, visual_bert.py:267-269; its parameters exist for checkpoint compatibility and receive no gradient, as there)."""

    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)
        self.seq_relationship = Linear(config.hidden_size, 2)
