"""BertVisioLinguisticEmbeddings with the reference's parameters (mmf/modules/embeddings.py:309-459)
and a single fused forward: text rows (gather + sum), visual rows (fp32 features -> bf16 MFMA GEMM
with bias / type / position added in the epilogue) written into one `[B, T+R, H]` buffer, LayerNorm,
dropout — all gfx950 kernels (mmf_amd.functional.VisioLinguisticEmbeddingsFn)."""
from copy import deepcopy
from typing import Optional

import torch
from torch import Tensor, nn

from mmf_amd import functional as Fn
from mmf_amd.modules.hf_layers import LayerNorm, Linear


class BertVisioLinguisticEmbeddings(nn.Module):
    def __init__(self, config, *args, **kwargs):
        super().__init__()
        H = config.hidden_size
        # HF BertEmbeddings members (embeddings.py:309-311 -> super().__init__)
        self.word_embeddings = nn.Embedding(config.vocab_size, H, padding_idx=getattr(config, "pad_token_id", 0))
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, H)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, H)
        self.LayerNorm = LayerNorm(H, eps=config.layer_norm_eps)
        self.dropout_prob = config.hidden_dropout_prob
        self.pad_idx = -1 if self.word_embeddings.padding_idx is None else int(self.word_embeddings.padding_idx)
        # visual members (:312-319)
        self.token_type_embeddings_visual = nn.Embedding(config.type_vocab_size, H)
        self.position_embeddings_visual = nn.Embedding(config.max_position_embeddings, H)
        self.projection = Linear(config.visual_embedding_dim, H)

    def initialize_visual_from_pretrained(self):
        """embeddings.py:321-327."""
        self.token_type_embeddings_visual.weight = nn.Parameter(
            deepcopy(self.token_type_embeddings.weight.data), requires_grad=True)
        self.position_embeddings_visual.weight = nn.Parameter(
            deepcopy(self.position_embeddings.weight.data), requires_grad=True)

    def forward(self, input_ids: Tensor, token_type_ids: Optional[Tensor] = None, visual_embeddings: Optional[Tensor] = None,
                visual_embeddings_type: Optional[Tensor] = None, image_text_alignment: Optional[Tensor] = None) -> Tensor:
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        pad = self.pad_idx
        return torch.ops.mmf_amd.visio_linguistic_embeddings(
            input_ids, token_type_ids, visual_embeddings, visual_embeddings_type,
            self.word_embeddings.weight, self.position_embeddings.weight, self.token_type_embeddings.weight,
            self.LayerNorm.weight, self.LayerNorm.bias, self.token_type_embeddings_visual.weight,
            self.position_embeddings_visual.weight, self.projection.weight, self.projection.bias,
            self.LayerNorm.eps, self.dropout_prob, self.training, pad, image_text_alignment)
