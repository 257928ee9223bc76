"""VisualBERT behind MMF's model API, running on the gfx950 kernels.

Mirrors mmf/models/visual_bert.py: `VisualBERTBase` (:43-157), `VisualBERTForPretraining` (:160-281),
`VisualBERTForClassification` (:284-404) and the registered `VisualBERT(BaseModel)` (:407-601) — same constructor arguments, same
`forward(sample_list) -> {"scores": [B, num_labels]}` contract, same parameter names/shapes (MMF zoo
checkpoints load unmodified: `model.bert.encoder.layer.3.attention.self.query.weight` ...).

MI355X-first differences, all internal:
  * activations are bf16 token-major tensors in HBM, statistics / softmax / accumulators fp32;
  * the whole embedding stage, each attention block and each feed-forward block is ONE autograd node
    made of hand-written HIP kernels (mmf_amd/functional.py);
  * the BertPooler is skipped when `pooler_strategy == "vqa"` (the reference computes it at :146 and
    discards it at :389-398); its parameters still exist and simply receive no gradient, exactly as
    in the reference.
"""
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from mmf_amd import functional as Fn
from mmf_amd import ops  # noqa: F401  (registers torch.ops.mmf_amd.*)
from mmf_amd.common.registry import registry
from mmf_amd.models.base_model import BaseModel
from mmf_amd.modules.embeddings import BertVisioLinguisticEmbeddings
from mmf_amd.modules.hf_layers import (
    BertConfig, BertEncoderJit, BertLayerJit, BertPooler, BertPredictionHeadTransform, BertPreTrainingHeads, Linear, init_bert_weights)
from mmf_amd.utils.configuration import to_container
from mmf_amd.utils.modeling import get_optimizer_parameters_for_bert


class VisualBERTBase(nn.Module):
    """visual_bert.py:43-157."""

    def __init__(self, config, visual_embedding_dim=512, embedding_strategy="plain", bypass_transformer=False,
                 output_attentions=False, output_hidden_states=False, skip_pooler=False):
        super().__init__()
        self.config = config
        config.visual_embedding_dim = visual_embedding_dim
        config.embedding_strategy = embedding_strategy
        config.bypass_transformer = bypass_transformer
        config.output_attentions = output_attentions
        config.output_hidden_states = output_hidden_states
        # `output_attentions: true` is accepted like the reference accepts it — and yields what the reference yields: VisualBERTBase.forward calls the encoder
        # WITHOUT asking for the layers' probabilities (visual_bert.py:143-157 `self.encoder(embedding_output, extended_attention_mask)`;
        # BertEncoderJit.forward collects them only when its own `output_attentions` argument is set, hf_layers.py:322-356), so
        # `attn_data_list = encoded_layers[1:]` — the `attention_weights` entry of the model output — is EMPTY there.  Nothing to materialise.
        self.embeddings = BertVisioLinguisticEmbeddings(config)
        self.encoder = BertEncoderJit(config)
        self.pooler = BertPooler(config)
        self.bypass_transformer = bypass_transformer
        if self.bypass_transformer:
            self.additional_layer = BertLayerJit(config)
        self.output_attentions = output_attentions
        self.output_hidden_states = output_hidden_states
        self.skip_pooler = skip_pooler
        self.init_weights()

    def _init_weights(self, module):
        init_bert_weights(module, self.config.initializer_range)

    def init_weights(self):
        self.apply(self._init_weights)

    @torch.jit.unused      # (eager only, like the reference's bypass model in practice: the joint layer is reached through autograd Functions)
    def _bypass_forward(self, embedding_output: Tensor, mask_add: Tensor, text_length: int) -> Tuple[Tensor, Tensor]:
        """visual_bert.py:116-141: the text rows alone go through the encoder (the mask slice `[:, :, :T, :T]` of the [B, 1, 1, S]
        additive mask keeps the broadcast query dimension and the first T keys), then ONE `additional_layer` sees text + regions."""
        B, S = mask_add.shape[0], mask_add.shape[1]
        text_embedding_output = embedding_output[:, :text_length, :]
        visual_part = embedding_output[:, text_length:, :]
        text_mask = mask_add[:, :text_length].contiguous().view(B, 1, 1, text_length)
        sequence_output = self.encoder(text_embedding_output, text_mask)[0]
        new_input = Fn.ConcatRowsFn.apply(sequence_output, visual_part)                      # torch.cat(dim=1), :135
        final_sequence_output = self.additional_layer(new_input, mask_add.view(B, 1, 1, S))[0]
        return final_sequence_output, self.pooler(final_sequence_output)

    def forward(self, input_ids: Tensor, attention_mask: Optional[Tensor] = None, token_type_ids: Optional[Tensor] = None,
                visual_embeddings: Optional[Tensor] = None, visual_embeddings_type: Optional[Tensor] = None,
                image_text_alignment: Optional[Tensor] = None, mask_add: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor], List[Tensor]]:
        """`mask_add` (optional, not in the reference's signature): the additive mask of `attention_mask` when the caller already has it
        (VisualBERT.forward builds it in the same launch as the masks themselves, torch.ops.mmf_amd.visual_masks)."""
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        # additive mask (1 - m) * -10000, visual_bert.py:94-106, built by a HIP kernel as fp32 [B, S]
        if mask_add is None:
            mask_add = torch.ops.mmf_amd.additive_mask(attention_mask)
        extended_attention_mask = mask_add.view(mask_add.shape[0], 1, 1, mask_add.shape[1])

        embedding_output = self.embeddings(input_ids, token_type_ids, visual_embeddings=visual_embeddings,
                                           visual_embeddings_type=visual_embeddings_type,
                                           image_text_alignment=image_text_alignment)
        hidden: List[Tensor] = []
        if self.bypass_transformer and visual_embeddings is not None:
            assert not self.output_hidden_states       # (not supported on the bypass path, visual_bert.py:121-123)
            final_sequence_output, bypass_pooled = self._bypass_forward(embedding_output, mask_add, input_ids.size(1))
            return final_sequence_output, bypass_pooled, hidden
        if torch.jit.is_scripting():
            sequence_output = self.encoder(embedding_output, extended_attention_mask)[0]
        else:
            encoded_layers = self.encoder(embedding_output, extended_attention_mask, output_hidden_states=self.output_hidden_states)
            sequence_output = encoded_layers[0]
        pooled_output: Optional[Tensor] = None
        if not self.skip_pooler:
            pooled_output = self.pooler(sequence_output)
        return sequence_output, pooled_output, hidden


class VisualBERTForPretraining(nn.Module):
    """visual_bert.py:160-281: masked-language-model pretraining over the joint text + region sequence."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.output_attentions = self.config.output_attentions
        self.output_hidden_states = self.config.output_hidden_states
        self.bert_model_name = self.config.get("bert_model_name", None)
        self.bert_config = BertConfig.from_dict(to_container(self.config))
        # (offline: same architecture as `bert_model_name`, weights arrive via load_state_dict — see VisualBERTForClassification)
        # The pooled output feeds only `cls.seq_relationship`, whose score is dropped (:267-269): the pooler runs only when
        # `output_hidden_states` asks for `pooled_output` in the output dict.
        self.bert = VisualBERTBase(
            self.bert_config, visual_embedding_dim=self.config.visual_embedding_dim,
            embedding_strategy=self.config.embedding_strategy, bypass_transformer=self.config.bypass_transformer,
            output_attentions=self.config.output_attentions, output_hidden_states=self.config.output_hidden_states,
            skip_pooler=not self.config.output_hidden_states)
        self.vocab_size = self.bert.config.vocab_size
        self.cls = BertPreTrainingHeads(self.bert.config)
        self.ignore_index = -1                      # nn.CrossEntropyLoss(ignore_index=-1), :215
        self.init_weights()

    def init_weights(self):
        if self.config.get("random_initialize", False) is False:
            if self.bert_model_name is None:
                self.bert.init_weights()
            self.cls.apply(self.bert._init_weights)
            self.tie_weights()

    def tie_weights(self):
        """:227-235 — the decoder shares the word-embedding Parameter (no clone: this module is not exported to TorchScript,
        the reference refuses the pretraining head in script mode, :597-598)."""
        self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight

    def forward(self, input_ids: Tensor, input_mask: Tensor, attention_mask: Optional[Tensor] = None,
                token_type_ids: Optional[Tensor] = None, visual_embeddings: Optional[Tensor] = None,
                visual_embeddings_type: Optional[Tensor] = None, image_text_alignment: Optional[Tensor] = None,
                masked_lm_labels: Optional[Tensor] = None, mask_add: Optional[Tensor] = None,
                pool_index: Optional[Tensor] = None) -> Dict[str, Tensor]:
        sequence_output, pooled_output, attention_weights = self.bert(input_ids, attention_mask, token_type_ids, visual_embeddings,
                                                                      visual_embeddings_type, image_text_alignment, mask_add)
        output_dict: Dict[str, Tensor] = {}
        if not torch.jit.is_scripting():
            if self.output_attentions:      # :258-259 (empty in the reference as well: see VisualBERTBase.__init__)
                output_dict["attention_weights"] = attention_weights
        if self.output_hidden_states:
            output_dict["sequence_output"] = sequence_output
            if pooled_output is not None:
                output_dict["pooled_output"] = pooled_output
        if masked_lm_labels is not None:
            heads = self.cls.predictions
            hidden = heads.transform(sequence_output)
            loss, logits = torch.ops.mmf_amd.masked_lm_head(hidden, heads.decoder.weight, heads.bias, masked_lm_labels,
                                                            self.ignore_index)
            output_dict["logits"] = logits
            output_dict["masked_lm_loss"] = loss
            output_dict["loss"] = loss
        return output_dict


class VisualBERTForClassification(nn.Module):
    """visual_bert.py:284-404."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.output_attentions = self.config.output_attentions
        self.output_hidden_states = self.config.output_hidden_states
        self.pooler_strategy = self.config.get("pooler_strategy", "default")
        self.bert_model_name = getattr(self.config, "bert_model_name", None)
        self.bert_config = BertConfig.from_dict(to_container(self.config))
        # The reference downloads `bert-base-uncased` when bert_model_name is set (visual_bert.py:309-320).
        # Offline we build the same architecture (BertConfig defaults == bert-base-uncased, overridden by any
        # BERT key present in the model config); real weights arrive via load_state_dict / an MMF checkpoint.
        self.bert = VisualBERTBase(
            self.bert_config, visual_embedding_dim=self.config.visual_embedding_dim,
            embedding_strategy=self.config.embedding_strategy, bypass_transformer=self.config.bypass_transformer,
            output_attentions=self.config.output_attentions, output_hidden_states=self.config.output_hidden_states,
            skip_pooler=(self.pooler_strategy == "vqa"))
        self.training_head_type = self.config.training_head_type
        self.num_labels = self.config.num_labels
        self.dropout_prob = self.bert.config.hidden_dropout_prob
        if self.training_head_type == "nlvr2":
            if self.pooler_strategy == "vqa":
                raise ValueError("training_head_type nlvr2 pairs the BertPooler outputs; pooler_strategy must not be 'vqa'")
            self.bert.config.hidden_size *= 2          # visual_bert.py:324-325
        self.classifier = nn.Sequential(
            BertPredictionHeadTransform(self.bert.config),
            Linear(self.bert.config.hidden_size, self.config.num_labels),
        )
        self.init_weights()

    def init_weights(self):
        if self.config.get("random_initialize", False) is False:
            self.classifier.apply(self.bert._init_weights)
        if "losses" in self.config and self.config.get("zerobias", False):
            for loss in self.config.losses:
                if "bce" in loss["type"]:
                    self.classifier[1].bias.data.fill_(self.config.biasfill)

    def forward(self, input_ids: Tensor, input_mask: Tensor, attention_mask: Optional[Tensor] = None,
                token_type_ids: Optional[Tensor] = None, visual_embeddings: Optional[Tensor] = None,
                visual_embeddings_type: Optional[Tensor] = None, image_text_alignment: Optional[Tensor] = None,
                masked_lm_labels: Optional[Tensor] = None, mask_add: Optional[Tensor] = None,
                pool_index: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """`mask_add` / `pool_index` (optional, not in the reference's signature): the additive attention mask and `input_mask.sum(1) - 2`
        when VisualBERT.forward has already computed them (one launch, torch.ops.mmf_amd.visual_masks)."""
        sequence_output, pooled_output, attention_weights = self.bert(input_ids, attention_mask, token_type_ids, visual_embeddings,
                                                                      visual_embeddings_type, image_text_alignment, mask_add)
        if self.training_head_type == "nlvr2":
            assert pooled_output is not None
            pooled_output = torch.ops.mmf_amd.pair_halves(pooled_output)       # 2B x H -> B x 2H, visual_bert.py:369-374
        output_dict: Dict[str, Tensor] = {}
        if not torch.jit.is_scripting():
            if self.output_attentions:      # :378-379 (empty in the reference as well: see VisualBERTBase.__init__)
                output_dict["attention_weights"] = attention_weights
        if self.output_hidden_states:
            output_dict["sequence_output"] = sequence_output
            if pooled_output is not None:
                output_dict["pooled_output"] = pooled_output
        if self.pooler_strategy == "vqa":
            # representation of the second-to-last text token (visual_bert.py:389-398) + dropout (:400)
            index_to_gather = pool_index if pool_index is not None else input_mask.sum(1) - 2
            pooled = torch.ops.mmf_amd.gather_rows(sequence_output, index_to_gather, self.dropout_prob, self.training)
        else:
            assert pooled_output is not None
            pooled = torch.ops.mmf_amd.dropout(pooled_output, self.dropout_prob, self.training)
        hidden = self.classifier[0](pooled)
        logits = self.classifier[1](hidden, True)
        output_dict["scores"] = logits.contiguous().view(-1, self.num_labels)
        return output_dict


@registry.register_model("visual_bert")
class VisualBERT(BaseModel):
    """visual_bert.py:407-601."""

    def __init__(self, config):
        super().__init__(config)
        self.config = config
        self.training_head_type = self.config.training_head_type

    @classmethod
    def config_path(cls):
        return "configs/models/visual_bert/pretrain.yaml"

    def build(self):
        if self.training_head_type == "pretraining":
            self.model = VisualBERTForPretraining(self.config)
        else:
            self.model = VisualBERTForClassification(self.config)
        if self.config.get("special_visual_initialize", False):
            self.model.bert.embeddings.initialize_visual_from_pretrained()
        if getattr(self.config, "freeze_base", False):
            for p in self.model.bert.parameters():
                p.requires_grad = False

    def get_optimizer_parameters(self, config):
        return get_optimizer_parameters_for_bert(self.model, config)

    @classmethod
    def format_state_key(cls, key):
        return (key.replace("bert.bert", "model.bert").replace("bert.cls", "model.cls")
                .replace("bert.classifier", "model.classifier"))

    # ---- input massaging, visual_bert.py:444-556 ------------------------------------------------
    def update_sample_list_based_on_head(self, sample_list: Dict[str, Tensor]) -> Dict[str, Tensor]:
        bert_input_ids, bert_input_mask = sample_list["input_ids"], sample_list["input_mask"]
        bert_input_type_ids = sample_list["segment_ids"]
        image_dim_variable: Optional[Tensor] = None
        if self.training_head_type == "nlvr2":          # visual_bert.py:490-514: text repeated, the two images stacked
            if not torch.jit.is_scripting():
                bert_input_ids = torch.cat([bert_input_ids, bert_input_ids])
                bert_input_mask = torch.cat([bert_input_mask, bert_input_mask])
                bert_input_type_ids = torch.cat([bert_input_type_ids, bert_input_type_ids])
                img0, img1 = sample_list.get("img0", None) or {}, sample_list.get("img1", None) or {}
                image_feat_variable = torch.cat([img0["image_feature_0"], img1["image_feature_0"]])
                d0 = (img0.get("image_info_0", None) or {}).get("max_features", None)
                d1 = (img1.get("image_info_0", None) or {}).get("max_features", None)
                image_dim_variable = torch.cat([d0, d1]) if d0 is not None and d1 is not None else None
            else:
                raise RuntimeError("nlvr2 head doesn't support scripting as of now")
        else:
            if not torch.jit.is_scripting():
                image_info = sample_list.get("image_info_0", None) or {}
                image_dim_variable = image_info.get("max_features", None)
                image_feat_variable = sample_list.get("image_feature_0", None)
            else:
                image_feat_variable = sample_list["image_feature_0"]      # (a Dict[str, Tensor] cannot nest image_info_0)
        sample_list["input_ids"], sample_list["input_mask"] = bert_input_ids, bert_input_mask
        sample_list["segment_ids"] = bert_input_type_ids
        if image_dim_variable is None:
            image_dim_variable = sample_list["image_feature_0"].new_full(
                size=(image_feat_variable.size(0), 1), fill_value=image_feat_variable.size(1))
        sample_list["visual_embeddings"] = image_feat_variable
        sample_list["image_dim"] = image_dim_variable
        sample_list["token_type_ids"] = sample_list["segment_ids"]
        return sample_list

    def add_custom_params(self, sample_list: Dict[str, Tensor]) -> Dict[str, Tensor]:
        visual_embeddings = sample_list["visual_embeddings"]
        image_dim = sample_list["image_dim"]
        if self.training_head_type == "pretraining":
            sample_list["masked_lm_labels"] = sample_list["lm_label_ids"]          # visual_bert.py:539-541
        # image_mask = arange(R) < image_dim (visual_bert.py:544-555) and, in the SAME launch, what add_post_flatten_params, VisualBERTBase and
        # the `vqa` pooling derive from the masks next: visual_embeddings_type (zeros), attention_mask (the concatenation), its additive
        # form and input_mask.sum(1) - 2 — one kernel instead of arange / compare / cast / zeros_like / cat / sum / subtract / additive mask
        image_mask, attention_mask, vtype, mask_add, pool_index = torch.ops.mmf_amd.visual_masks(
            sample_list["input_mask"], image_dim, visual_embeddings.size(-2))
        sample_list["image_mask"] = image_mask
        sample_list["_fused_attention_mask"] = attention_mask
        sample_list["_fused_visual_embeddings_type"] = vtype
        sample_list["_fused_mask_add"] = mask_add
        sample_list["_fused_pool_index"] = pool_index
        return sample_list

    def add_post_flatten_params(self, sample_list: Dict[str, Tensor]) -> Dict[str, Tensor]:
        if "_fused_attention_mask" in sample_list:      # computed with the image mask (add_custom_params)
            sample_list["visual_embeddings_type"] = sample_list["_fused_visual_embeddings_type"]
            sample_list["attention_mask"] = sample_list["_fused_attention_mask"]
        else:
            sample_list["visual_embeddings_type"] = torch.zeros_like(sample_list["image_mask"])
            sample_list["attention_mask"] = torch.cat((sample_list["input_mask"], sample_list["image_mask"]), dim=-1)
        if self.training_head_type == "pretraining":
            # visual_bert.py:455-465: labels over the joint sequence, -1 (ignored) on every visual position
            lm = sample_list["masked_lm_labels"]
            assert lm.dim() == 2 and lm.size(-1) == sample_list["input_mask"].size(-1)
            new_lm_labels = torch.full_like(sample_list["attention_mask"], -1)
            new_lm_labels[: lm.size(0), : lm.size(1)] = lm
            sample_list["masked_lm_labels"] = new_lm_labels
        return sample_list

    def forward(self, sample_list: Dict[str, Tensor]) -> Dict[str, Tensor]:
        if torch.jit.is_scripting():
            assert "image_feature_0" in sample_list, "Key 'image_feature_0' is required in TorchScript model"
        sample_list = self.update_sample_list_based_on_head(sample_list)
        sample_list = self.add_custom_params(sample_list)
        sample_list = self.add_post_flatten_params(sample_list)
        image_text_alignment: Optional[Tensor] = None
        masked_lm_labels: Optional[Tensor] = None
        if "image_text_alignment" in sample_list:
            image_text_alignment = sample_list["image_text_alignment"]
        if "masked_lm_labels" in sample_list:
            masked_lm_labels = sample_list["masked_lm_labels"]
        mask_add: Optional[Tensor] = None
        pool_index: Optional[Tensor] = None
        if "_fused_mask_add" in sample_list:
            mask_add = sample_list["_fused_mask_add"]
            pool_index = sample_list["_fused_pool_index"]
        output_dict = self.model(
            sample_list["input_ids"], sample_list["input_mask"], sample_list["attention_mask"],
            sample_list["token_type_ids"], sample_list["visual_embeddings"], sample_list["visual_embeddings_type"],
            image_text_alignment, masked_lm_labels, mask_add, pool_index)
        if self.training_head_type == "pretraining":                               # visual_bert.py:588-598
            if not torch.jit.is_scripting():
                loss_key = "{}/{}".format(sample_list["dataset_name"], sample_list["dataset_type"])
                output_dict["losses"] = {loss_key + "/masked_lm_loss": output_dict.pop("masked_lm_loss")}
            else:
                raise RuntimeError("Pretraining head can't be used in script mode.")
        return output_dict
