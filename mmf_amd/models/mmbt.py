"""MMBT behind MMF's model API, on the gfx950 kernels (BASELINE.json configs[0]; SURVEY.md §8 "next" row).

Mirrors mmf/models/mmbt.py: `MMBTConfig` (:42-64), `ModalEmbeddings` (:67-129), `MMBTModel` (:132-324),
`MMBTBase` (:327-444), `MMBTForClassification` (:526-563) and the registered `MMBT(BaseModel)` (:566-658):
same constructor arguments, same `forward(sample_list) -> {"scores": [B, num_labels]}`, same parameter tree
(`model.bert.mmbt.transformer.*`, `model.bert.mmbt.modal_encoder.*`, `model.classifier.*`), including the
modal encoder's ALIASES of the text embedding tables and LayerNorm (:78-82), so MMF checkpoints load unmodified.

Scope: `direct_features_input: true` — the modality arrives as pre-extracted `[B, N, modal_hidden_size]`
features and the modal encoder is the identity.  The CNN / detectron encoders that produce such features in the
reference (mmf/modules/encoders.py) sit upstream of the hot path and are out of scope; asking
for one raises.  The pretraining head (:447-523) is not built.

MI355X-first internals: the modal block (start token, projected features, end token) and the text block are
written into ONE `[B, L+T, H]` bf16 buffer and normalised by one LayerNorm kernel (`functional.MMBTEmbeddingsFn`)
instead of two embedding modules + `torch.cat`; the pooler's tanh lives in the GEMM epilogue.
"""
import torch
from torch import nn

from mmf_amd import fp32_path as F32P
from mmf_amd import fp32_train as F32T
from mmf_amd import functional as Fn
from mmf_amd.common.registry import registry
from mmf_amd.models.base_model import BaseModel
from mmf_amd.modules.encoders import MultiModalEncoderBase
from mmf_amd.modules.hf_layers import BertPredictionHeadTransform, BertPreTrainingHeads, Linear, init_bert_weights
from mmf_amd.utils.configuration import to_container
from mmf_amd.utils.modeling import get_optimizer_parameters_for_bert


class MMBTConfig:
    """mmbt.py:42-64 — a view of the transformer config plus the modal width."""

    def __init__(self, config, num_labels=None, modal_hidden_size=2048):
        self.__dict__ = config.__dict__
        self.modal_hidden_size = modal_hidden_size
        if num_labels:
            self.num_labels = num_labels


class ModalEmbeddings(nn.Module):
    """mmbt.py:67-129.  Holds `proj_embeddings` and aliases of the transformer's embedding tables."""

    def __init__(self, config, encoder, embeddings):
        super().__init__()
        self.config = config
        self.encoder = encoder
        self.proj_embeddings = Linear(config.modal_hidden_size, config.hidden_size)
        self.position_embeddings = embeddings.position_embeddings
        self.token_type_embeddings = embeddings.token_type_embeddings
        self.word_embeddings = embeddings.word_embeddings
        self.LayerNorm = embeddings.LayerNorm
        self.dropout_prob = config.hidden_dropout_prob


class MMBTModel(nn.Module):
    """mmbt.py:132-324.  `config.is_decoder`: the padding mask times a causal mask over the modal + text positions (:260-272), handed to the
    attention kernels as a materialised per-(query, key) mask (mmf_attn_desc.mask_query_stride); `encoder_hidden_states` are accepted and
    unused, as in the reference (its BertLayerJit.forward never calls the `crossattention` block it builds, hf_layers.py:268-292)."""

    def __init__(self, config, transformer, encoder):
        super().__init__()
        self.is_decoder = bool(getattr(config, "is_decoder", False))
        self.num_hidden_layers = config.num_hidden_layers
        self.transformer = transformer
        self.modal_encoder = ModalEmbeddings(config, encoder, transformer.embeddings)

    def forward(self, input_modal, input_ids, modal_start_tokens=None, modal_end_tokens=None, attention_mask=None,
                token_type_ids=None, modal_token_type_ids=None, position_ids=None, modal_position_ids=None, head_mask=None,
                inputs_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None):
        if position_ids is not None or modal_position_ids is not None or inputs_embeds is not None:
            raise NotImplementedError("explicit position ids / inputs_embeds are not on the built MMBT path")
        # (`head_mask` is accepted and unused, as in the reference: MMBTModel.forward never hands it to the encoder, mmbt.py:302-307)
        input_modal = self.modal_encoder.encoder(input_modal)
        if input_modal.dim() == 2:
            input_modal = input_modal.unsqueeze(1)
        B, N = input_modal.shape[0], input_modal.shape[1]
        L = N + (modal_start_tokens is not None) + (modal_end_tokens is not None)
        if token_type_ids is None:
            token_type_ids = torch.ones_like(input_ids)                                     # :213-216
        if modal_token_type_ids is None:
            modal_type = 0                                                                  # :117-122
        elif isinstance(modal_token_type_ids, int) or (isinstance(modal_token_type_ids, torch.Tensor)
                                                       and modal_token_type_ids.numel() == 1):
            modal_type = modal_token_type_ids          # (a one-element device tensor stays on the device)
        else:                                          # per-position ids [B, L] (:117-127 looks every one up): functional.mmbt_modal_types
            if tuple(modal_token_type_ids.shape) != (B, L):
                raise ValueError("modal_token_type_ids must be [batch, %d] (start token, %d modal rows, end token), got %s"
                                 % (L, N, tuple(modal_token_type_ids.shape)))
            modal_type = modal_token_type_ids
        emb, me = self.transformer.embeddings, self.modal_encoder
        if F32T.active():        # mmf_amd.fp32_training(): fp32 forward + backward
            hidden = F32T.mmbt_embeddings(
                input_modal, input_ids, modal_start_tokens, modal_end_tokens, token_type_ids, modal_type,
                emb.word_embeddings.weight, emb.position_embeddings.weight, emb.token_type_embeddings.weight,
                emb.LayerNorm.weight, emb.LayerNorm.bias, me.proj_embeddings.weight, me.proj_embeddings.bias, emb.LayerNorm.eps,
                emb.dropout_prob, self.training, emb.word_embeddings.padding_idx)
        elif F32P.active():        # fp32-accurate forward (mmf_amd.fp32_inference()): same buffer layout, fp32 kernels
            F32P.check_no_dropout(emb.dropout_prob, self.training)
            hidden = F32P.mmbt_embeddings(
                input_modal, input_ids, modal_start_tokens, modal_end_tokens, token_type_ids, modal_type,
                emb.word_embeddings.weight, emb.position_embeddings.weight, emb.token_type_embeddings.weight,
                emb.LayerNorm.weight, emb.LayerNorm.bias, me.proj_embeddings.weight, me.proj_embeddings.bias, emb.LayerNorm.eps)
        else:
            hidden = Fn.MMBTEmbeddingsFn.apply(
                input_modal, input_ids, modal_start_tokens, modal_end_tokens, token_type_ids, modal_type,
                emb.word_embeddings.weight, emb.position_embeddings.weight, emb.token_type_embeddings.weight,
                emb.LayerNorm.weight, emb.LayerNorm.bias, me.proj_embeddings.weight, me.proj_embeddings.bias,
                Fn.shadows.get(me.proj_embeddings.weight), emb.LayerNorm.eps, Fn.make_drop(emb.dropout_prob, self.training),
                emb.word_embeddings.padding_idx)
        S = hidden.shape[1]
        dev = hidden.device
        if attention_mask is None:
            am = torch.ones(B, S, dtype=torch.int64, device=dev)                            # :228-229
        else:
            am = torch.cat([torch.ones(B, L, dtype=torch.int64, device=dev), attention_mask.long()], dim=1)  # :232-238
        if self.is_decoder:                                                                 # :260-272, then (1 - m) * -10000 (:283)
            seq_ids = torch.arange(S, device=dev)
            causal = seq_ids[None, :] <= seq_ids[:, None]                                   # [query, key]: key <= query
            ext = (causal[None, :, :] & (am[:, None, :] != 0)).to(torch.float32)
            encoder_outputs = self.transformer.encoder(hidden, ((1.0 - ext) * -10000.0).view(B, 1, S, S))
            sequence_output = encoder_outputs[0]
            return sequence_output, self.transformer.pooler(sequence_output), encoder_outputs[1:]
        mask_add = torch.empty(B, S, dtype=torch.float32, device=dev)
        Fn.nat.make_additive_mask(am.contiguous(), mask_add)                                # (1 - m) * -10000, :283
        encoder_outputs = self.transformer.encoder(hidden, mask_add.view(B, 1, 1, S))
        sequence_output = encoder_outputs[0]
        pooled_output = self.transformer.pooler(sequence_output)
        # :302-316 — `encoder_outputs[1:]`: the reference calls the (JIT-replaced) encoder without its `output_hidden_states` / `output_attentions` arguments
        # (hf_layers.py:317-356 collects them only when the ARGUMENTS are set), so this tail is empty there too, whatever the config flags say
        return sequence_output, pooled_output, encoder_outputs[1:]

    def get_input_embeddings(self):
        return self.transformer.embeddings.word_embeddings


class MMBTBase(MultiModalEncoderBase):
    """mmbt.py:327-444 on `MultiModalEncoderBase` (mmf/modules/encoders.py:588-646): the text encoder is the registered
    `"transformer"` encoder (`BertModelJit` on the HIP kernels), the modal encoder one of the image-FEATURE encoders (identity, the
    trainable fc7 layer `finetune_faster_rcnn_fpn_fc7` of projects/hateful_memes/configs/mmbt/with_features.yaml, a linear projection);
    raw-image CNN encoders raise (out of scope)."""

    def __init__(self, config, *args, **kwargs):
        super().__init__(config, *args, **kwargs)

    def build(self):
        encoders = self._build_encoders(self.config)
        text_encoder, modal_encoder = encoders[0], encoders[1]
        if text_encoder is None or not hasattr(text_encoder, "config"):
            raise NotImplementedError("MMBT needs a transformer text encoder (text_encoder.type: transformer)")
        if modal_encoder is None:
            modal_encoder = nn.Identity()
        self._encoder_config = text_encoder.config
        self._mmbt_config = MMBTConfig(self._encoder_config, num_labels=self.config.num_labels,
                                       modal_hidden_size=self.config.modal_hidden_size)
        self.use_modal_start_token = self.config.use_modal_start_token
        self.use_modal_end_token = self.config.use_modal_end_token
        te_params = (self.config.text_encoder.get("params", None) or {}) if self.config.get("text_encoder", None) else {}
        self.num_max_segment = te_params.get("num_segments", 2)                       # mmbt.py:345
        self.mmbt = MMBTModel(self._mmbt_config, text_encoder, modal_encoder)

    @property
    def encoder_config(self):
        return self._encoder_config

    def extract_modal_end_token(self, sample_list):
        """mmbt.py:346-363: the last real text token becomes the modal end token; text shifts left by one."""
        gather_index = sample_list["input_mask"].sum(1, keepdim=True) - 1
        modal_end_token = torch.gather(sample_list["input_ids"], 1, gather_index).squeeze(1).clone().detach()
        batch_size = sample_list["input_ids"].size(0)
        device = sample_list["input_ids"].device
        sample_list["input_ids"] = torch.cat([sample_list["input_ids"][:, 1:], sample_list["input_ids"][:, -1:]], dim=1)
        sample_list["input_mask"] = torch.cat(
            [sample_list["input_mask"][:, 1:], torch.zeros([batch_size, 1], dtype=torch.long, device=device)], dim=1)
        return modal_end_token

    def forward(self, sample_list):
        if self._is_direct_features_input:
            input_modal = sample_list["input_modal"] if "input_modal" in sample_list else sample_list["image_feature_0"]
        else:
            input_modal = sample_list["image"]
        modal_start_token = None
        if self.use_modal_start_token:
            modal_start_token = sample_list["input_ids"][:, 0].clone().detach()
        modal_end_token = None
        if self.use_modal_end_token:
            modal_end_token = self.extract_modal_end_token(sample_list)
        if "modal_token_type_ids" in sample_list:
            modal_token_type_ids = sample_list["modal_token_type_ids"]
        else:
            # mmbt.py:385-410 picks the modal block's segment id from the range of the text segment ids.  The reference compares
            # device tensors in Python (`if max_id == min_id`), a host read-back per step; the same decision table evaluated
            # on the device (no synchronisation, so the whole step can be captured in a hipGraph):
            #   max == min : 1 if max == 0 else 0          max != min : max_segment if max != max_segment else 0
            segment_ids = sample_list["segment_ids"]
            max_id, min_id = segment_ids.max(), segment_ids.min()
            max_segment = self.num_max_segment - 1
            modal_token_type_ids = torch.where(max_id == min_id, (max_id == 0).to(max_id.dtype),
                                               torch.where(max_id != max_segment, torch.full_like(max_id, max_segment),
                                                           torch.zeros_like(max_id))).reshape(1)
        if input_modal.dim() == 2:
            input_modal = input_modal.unsqueeze(dim=1)
        return self.mmbt(input_modal, input_ids=sample_list["input_ids"], modal_start_tokens=modal_start_token,
                         modal_end_tokens=modal_end_token, attention_mask=sample_list["input_mask"],
                         token_type_ids=sample_list["segment_ids"], modal_token_type_ids=modal_token_type_ids)


class MMBTForPreTraining(nn.Module):
    """mmbt.py:447-523: masked-language-model pretraining; the text positions are the LAST T rows of the joint sequence
    (modal block first), `cls` = HF BertPreTrainingHeads with the decoder tied to the word embeddings (:467-476)."""

    def __init__(self, config, *args, **kwargs):
        super().__init__()
        self.config = config
        self.bert = MMBTBase(config, *args, **kwargs)
        self.encoder_config = self.bert.encoder_config
        # (offline: the architecture of `bert_model_name`; weights arrive through load_state_dict / an MMF checkpoint)
        self.cls = BertPreTrainingHeads(self.encoder_config)
        self.cls.apply(lambda m: init_bert_weights(m, self.encoder_config.initializer_range))
        self.ignore_index = -1                                          # nn.CrossEntropyLoss(ignore_index=-1), :464
        self.tie_weights()

    def tie_weights(self):
        self.cls.predictions.decoder.weight = self.bert.mmbt.transformer.embeddings.word_embeddings.weight

    def forward(self, sample_list):
        module_output = self.bert(sample_list)
        sequence_output = module_output[0]
        output = {}
        if self.encoder_config.output_hidden_states or self.encoder_config.output_attentions:      # :486-490 (an empty tail in the reference as well: see MMBTModel.forward)
            output["extras"] = module_output[2:]
        loss_key = "{}/{}".format(sample_list["dataset_name"], sample_list["dataset_type"])
        lm_label_ids = sample_list["lm_label_ids"] if "lm_label_ids" in sample_list else None
        if lm_label_ids is not None:
            # :494-506 scores only the last T (text) positions: the same mean as CrossEntropyLoss over ALL positions with the
            # modal block's labels set to ignore_index, which is what the fused decoder + loss takes
            B, S = sequence_output.shape[0], sequence_output.shape[1]
            T = lm_label_ids.size(1)
            labels = torch.full((B, S), self.ignore_index, dtype=torch.int64, device=lm_label_ids.device)
            labels[:, S - T:] = lm_label_ids
            heads = self.cls.predictions
            hidden = heads.transform(sequence_output)
            loss, logits = torch.ops.mmf_amd.masked_lm_head(hidden, heads.decoder.weight, heads.bias, labels, self.ignore_index)
            output["logits"] = logits
            output["losses"] = {loss_key + "/masked_lm_loss": loss}
        if "image_text_alignment" in sample_list and sample_list["image_text_alignment"] is not None:
            # :509-518 hands CrossEntropyLoss the flattened [2B] next-sentence scores with a [B] target, which torch rejects for
            # B > 1: that branch cannot run in the reference either
            raise NotImplementedError("MMBTForPreTraining alignment loss (mmbt.py:509-518) is not built")
        return output


class MMBTForClassification(nn.Module):
    """mmbt.py:526-563."""

    def __init__(self, config, *args, **kwargs):
        super().__init__()
        self.config = config
        self.bert = MMBTBase(config, *args, **kwargs)
        self.encoder_config = self.bert.encoder_config
        self.num_labels = self.config.num_labels
        self.output_hidden_states = self.encoder_config.output_hidden_states
        self.output_attentions = self.encoder_config.output_attentions
        self.fused_feature_only = self.config.get("fused_feature_only", False)
        self.dropout_prob = self.encoder_config.hidden_dropout_prob
        self.classifier = nn.Sequential(
            BertPredictionHeadTransform(self.encoder_config),
            Linear(self.encoder_config.hidden_size, self.config.num_labels),
        )
        self.classifier.apply(lambda m: init_bert_weights(m, self.encoder_config.initializer_range))

    def forward(self, sample_list):
        module_output = self.bert(sample_list)
        pooled_output = module_output[1]
        output = {}
        if self.output_hidden_states or self.output_attentions:      # :545-547 (an empty tail in the reference as well: see MMBTModel.forward)
            output["extras"] = module_output[2:]
        pooled_output = torch.ops.mmf_amd.dropout(pooled_output, self.dropout_prob, self.training)
        if self.fused_feature_only:
            output["fused_feature"] = self.classifier[0](pooled_output)
            return output
        hidden = self.classifier[0](pooled_output)
        logits = self.classifier[1](hidden, out_f32=True)
        output["scores"] = logits.contiguous().view(-1, self.num_labels)
        return output


@registry.register_model("mmbt")
class MMBT(BaseModel):
    """mmbt.py:566-658."""

    def __init__(self, config, *args, **kwargs):
        super().__init__(config)

    @classmethod
    def config_path(cls):
        return "configs/models/mmbt/pretrain.yaml"

    def build(self):
        if self.config.get("training_head_type", "pretraining") == "pretraining":
            self.model = MMBTForPreTraining(self.config)
        else:
            self.model = MMBTForClassification(self.config)
        if self.config.get("freeze_complete_base", False) or self.config.get("freeze_text", False):
            for p in self.model.bert.mmbt.transformer.parameters():
                p.requires_grad = False
        if self.config.get("freeze_complete_base", False) or self.config.get("freeze_modal", False):
            for p in self.model.bert.mmbt.modal_encoder.parameters():
                p.requires_grad = False

    @classmethod
    def format_state_key(cls, key):
        return (key.replace("base.bert", "model.bert").replace("base.cls", "model.cls")
                .replace("base.classifier", "model.classifier"))

    def forward(self, sample_list):
        return self.model(sample_list)

    def get_optimizer_parameters(self, config):
        return get_optimizer_parameters_for_bert(self.model, config)
