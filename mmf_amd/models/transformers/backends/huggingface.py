"""The `huggingface` transformer backend of MMF Transformer on the HIP kernels
(mmf/models/transformers/backends/huggingface.py:19-235): `HuggingfaceEmbeddings` (one embedding stack per modality,
concatenated) + `HuggingfaceBackend` (BERT encoder).  Same parameter tree as the reference — the text modality aliases
the transformer's word table and LayerNorm (:106-109), every modality owns a copy of the position table (:111-114) —
so MMFT checkpoints load unmodified.

Only BERT bases are built (`transformer_base: bert-*` or None); the reference's `AutoModel` fallback for other
architectures (:183-186) raises here.  There is no network: the architecture comes from `transformer_base`'s known
config (bert-base-uncased / bert-large-uncased) overridden by the model config, weights from a checkpoint."""
from copy import deepcopy

import torch
from torch import nn

from mmf_amd import fp32_path as F32P
from mmf_amd import fp32_train as F32T
from mmf_amd import functional as Fn
from mmf_amd.common.registry import registry
from mmf_amd.models.transformers.base import BaseTransformerBackend
from mmf_amd.modules.hf_layers import BertConfig, BertModelJit, Dropout, LayerNorm, Linear
from mmf_amd.utils.configuration import to_container

_KNOWN_BASES = {
    None: {},
    "bert-base-uncased": {},
    "bert-large-uncased": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096),
}


class HuggingfaceEmbeddings(nn.Module):
    """huggingface.py:19-159."""

    def __init__(self, model_config, transformer_config, transformer, *args, **kwargs):
        super().__init__()
        self.model_config = model_config
        self.transformer_config = transformer_config
        self.token_embeddings = nn.ModuleList()
        self.pos_embeddings = nn.ModuleList()
        self.layer_norms = nn.ModuleList()
        self.dropouts = nn.ModuleList()
        self.modality_keys = []
        self.modality_is_text = []
        self.build_layers()
        self.init_weights(transformer)
        assert (len(self.token_embeddings) == len(self.pos_embeddings) == len(self.layer_norms) == len(self.dropouts)
                == len(self.model_config.modalities))

    def build_layers(self):
        tc = self.transformer_config
        for modality in self.model_config.modalities:
            self.modality_keys.append(modality["key"])
            eps = modality.get("layer_norm_eps", tc.layer_norm_eps)
            position_dim = modality.get("position_dim", tc.max_position_embeddings)
            p = modality.get("hidden_dropout_prob", tc.hidden_dropout_prob)
            is_text = modality["type"] == "text" and modality.get("consume_raw", True)
            self.modality_is_text.append(is_text)
            if is_text:
                self.token_embeddings.append(nn.Embedding(tc.vocab_size, tc.hidden_size, padding_idx=tc.pad_token_id))
            else:
                self.token_embeddings.append(nn.Sequential(Linear(modality["embedding_dim"], tc.hidden_size),
                                                           LayerNorm(tc.hidden_size, eps=eps)))
            self.pos_embeddings.append(nn.Embedding(position_dim, tc.hidden_size))
            self.layer_norms.append(LayerNorm(tc.hidden_size, eps=eps))
            self.dropouts.append(Dropout(p))
        self.token_type_embeddings = nn.Embedding(len(self.model_config.modalities), tc.hidden_size)

    def init_weights(self, transformer):
        """huggingface.py:104-129."""
        for idx, modality in enumerate(self.model_config.modalities):
            if modality["type"] == "text":
                self.token_embeddings[idx] = transformer.embeddings.word_embeddings
                self.layer_norms[idx] = transformer.embeddings.LayerNorm
            self.pos_embeddings[idx].weight = nn.Parameter(
                deepcopy(transformer.embeddings.position_embeddings.weight.data), requires_grad=True)
        n_type = self.transformer_config.type_vocab_size
        src = transformer.embeddings.token_type_embeddings.weight
        self.token_type_embeddings.weight.data[:n_type].copy_(src.data[: self.token_type_embeddings.weight.shape[0]])
        for idx in range(n_type, len(self.model_config.modalities)):
            self.token_type_embeddings.weight.data[idx].copy_(src.data.mean(dim=0))
            self.token_type_embeddings.weight.data[idx] += torch.normal(
                self.model_config.get("token_noise_mean", 0.0), self.model_config.get("token_noise_std", 0.01),
                size=self.token_type_embeddings.weight.data[idx].size())

    def forward(self, tokens_ids, position_ids, segment_ids):
        blocks = []
        for idx, key in enumerate(self.modality_keys):
            tok, pos_emb, ln, dropout = self.token_embeddings[idx], self.pos_embeddings[idx], self.layer_norms[idx], self.dropouts[idx]
            x = tokens_ids[key]
            pos_w = pos_emb.weight if key in position_ids else None
            seg = segment_ids.get(key, None)
            if self.modality_is_text[idx]:
                if pos_w is None:
                    raise NotImplementedError("a text modality without position ids is not on the fused embedding path")
                # word + position + type -> LayerNorm -> dropout, one fused stage (functional.VisioLinguisticEmbeddingsFn)
                typ_w = self.token_type_embeddings.weight
                if seg is None:
                    seg, typ_w = torch.zeros_like(x), torch.zeros_like(self.token_type_embeddings.weight)
                z = typ_w.new_zeros(1, typ_w.shape[1])
                if F32T.active():     # mmf_amd.fp32_training(): fp32 forward + backward
                    blocks.append(F32T.visio_linguistic_embeddings(x, seg, None, None, tok.weight, pos_w, typ_w, ln.weight, ln.bias, None, None,
                                                                   None, None, ln.eps, dropout.p, self.training, tok.padding_idx))
                    continue
                if F32P.active():     # fp32-accurate forward (mmf_amd.fp32_inference())
                    F32P.check_no_dropout(dropout.p, self.training)
                    blocks.append(F32P.visio_linguistic_embeddings(x, seg, None, None, tok.weight, pos_w, typ_w, ln.weight, ln.bias,
                                                                   z, z, z, z, ln.eps))
                    continue
                blocks.append(Fn.VisioLinguisticEmbeddingsFn.apply(
                    x, seg, None, None, tok.weight, pos_w, typ_w, ln.weight, ln.bias, z, z, z, z, None, ln.eps,
                    Fn.make_drop(dropout.p, self.training), tok.padding_idx))
                continue
            # Linear -> LayerNorm (the modality's token embedding), + position + type, LayerNorm, dropout
            h = tok[1](tok[0](x))
            typ_w = self.token_type_embeddings.weight if seg is not None else None
            if F32T.active():
                h = F32T.add_pos_type(h, seg, pos_w, typ_w)
            else:
                h = F32P.add_pos_type(h, seg, pos_w, typ_w) if F32P.active() else Fn.AddPosTypeFn.apply(h, seg, pos_w, typ_w)
            blocks.append(dropout(ln(h)))
        if len(blocks) == 1:
            return blocks[0]
        if F32T.active():
            return F32T.concat_rows(*blocks)
        return F32P.concat_rows(*blocks) if F32P.active() else Fn.ConcatRowsFn.apply(*blocks)


@registry.register_transformer_backend("huggingface")
class HuggingfaceBackend(BaseTransformerBackend):
    """huggingface.py:162-235."""

    def build_transformer_config(self):
        base = self.config.get("transformer_base", None)
        if base not in _KNOWN_BASES:
            raise NotImplementedError("transformer_base=%r: only BERT bases (%s) are built" % (base, sorted(k for k in _KNOWN_BASES if k)))
        d = dict(_KNOWN_BASES[base])
        d.update({k: v for k, v in to_container(self.config).items() if not isinstance(v, (dict, list))})
        self.transformer_config = BertConfig.from_dict(d)

    def build_transformer_base(self):
        self.transformer = BertModelJit(self.transformer_config)

    def build_embeddings(self):
        self.embeddings = HuggingfaceEmbeddings(self.config, self.transformer_config, self.transformer)

    def get_config(self):
        return self.transformer_config

    def generate_embeddings(self, tokens_ids, position_ids, segment_ids, attention_mask):
        return self.embeddings(tokens_ids=tokens_ids, position_ids=position_ids, segment_ids=segment_ids)

    def generate_attention_mask(self, masks):
        """(1 - cat(masks)) * -10000 as [B, 1, 1, S] (huggingface.py:204-210), built by the HIP mask kernel."""
        attention_mask = torch.cat([m.long() for m in masks], dim=-1).contiguous()
        mask_add = torch.empty(attention_mask.shape, dtype=torch.float32, device=attention_mask.device)
        Fn.nat.make_additive_mask(attention_mask, mask_add)
        return mask_add.view(attention_mask.shape[0], 1, 1, attention_mask.shape[1])

    def generate_encoded_layers(self, embedding, attention_mask):
        encoded_layers = self.transformer.encoder(embedding, attention_mask)
        return encoded_layers[-1], encoded_layers[0]
