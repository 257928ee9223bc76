"""UNITER (classification) behind MMF's model API on the gfx950 kernels (BASELINE.json configs[3]).

Mirrors mmf/models/uniter.py: `UNITERImageEmbeddings` (:45-87), `UNITERModelBase` (:90-246), `_infer_with_heads` (:249-275),
`UNITERForClassification` (:278-347) and the registered `UNITER(BaseModel)` (:621-773): same constructor arguments and config
keys (`heads`, `losses`, `tasks`, `img_dim`, `hidden_size`, `hidden_dropout_prob`, `text_embeddings`, `encoder`), same
`forward(sample_list) -> {"losses", "scores"}`, and the reference's parameter tree (`uniter.uniter.text_embeddings.*`,
`.img_embeddings.*`, `.encoder.layer.*`, `.pooler.*`, `uniter.heads.<task>.*`).

On the device: text rows are one fused embedding stage; region features + mask-embedding row are written once as bf16 and
projected by the MFMA GEMM; the 7-d box geometry goes through the same GEMM zero-padded to 8 columns; the three
LayerNorms, the sums and the concat are the row kernels of mmf_amd/csrc/rowops.hip; the trunk is the BERT encoder the
VisualBERT path runs on; heads come from the transformer-head registry (`mlp`).

`do_pretraining`: `UNITERForPretraining` (:350-618) for the tasks whose heads are built — mlm, itm, mrc (mmf_amd/models/transformers/
heads); mrfr (tied regression + MSE) and wra (optimal-transport alignment, 50 IPOT steps per sample in LDS) since round 3.  One task per batch, drawn like the
reference does; the random region masks come from the same numpy / random calls as the reference's `_get_img_mask`."""
import copy
import random
from collections import namedtuple

import numpy as np
import torch
from torch import nn

from mmf_amd import functional as Fn
from mmf_amd import fp32_path as F32P
from mmf_amd import fp32_train as F32T
from mmf_amd.common.registry import registry
from mmf_amd.models.base_model import BaseModel
from mmf_amd.models.transformers.heads import itm as _itm_head  # noqa: F401  (registers "itm")
from mmf_amd.models.transformers.heads import mlm as _mlm_head  # noqa: F401  (registers "mlm")
from mmf_amd.models.transformers.heads import mlp as _mlp_head  # noqa: F401  (registers "mlp")
from mmf_amd.models.transformers.heads import mrc as _mrc_head  # noqa: F401  (registers "mrc")
from mmf_amd.models.transformers.heads import mrfr as _mrfr_head  # noqa: F401  (registers "mrfr")
from mmf_amd.models.transformers.heads import wra as _wra_head  # noqa: F401  (registers "wra")
from mmf_amd.modules.hf_layers import (
    BertConfig, BertEmbeddingsJit, BertEncoderJit, BertPooler, Dropout, LayerNorm, Linear, init_bert_weights)
from mmf_amd.modules.losses import MMFLoss
from mmf_amd.utils.configuration import Config, to_container

_KNOWN_BASES = {None: {}, "bert-base-uncased": {},
                "bert-large-uncased": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)}


def _bert_config(bert_model_name, overrides):
    if bert_model_name not in _KNOWN_BASES:
        raise NotImplementedError("bert_model_name=%r: only BERT bases are built" % (bert_model_name,))
    d = dict(_KNOWN_BASES[bert_model_name])
    params = to_container(overrides or {})
    params = dict(params.get("params", params) or {})
    for k in ("bert_model_name", "type", "name", "random_init"):
        params.pop(k, None)
    d.update({k: v for k, v in params.items() if not isinstance(v, (dict, list))})
    return BertConfig.from_dict(d)


TransformerOutput = namedtuple("TransformerOutput", ["final_layer", "hidden_layers"])


class UNITERImageEmbeddings(nn.Module):
    """uniter.py:45-87."""

    def __init__(self, img_dim=2048, hidden_size=768, eps=1e-12, hidden_dropout_prob=0, pos_dim=7):
        super().__init__()
        self.img_linear = Linear(img_dim, hidden_size)
        self.img_layer_norm = LayerNorm(hidden_size, eps=eps)
        self.pos_layer_norm = LayerNorm(hidden_size, eps=eps)
        self.pos_linear = Linear(pos_dim, hidden_size)
        self.mask_embedding = nn.Embedding(2, img_dim, padding_idx=0)
        self.final_layer_norm = LayerNorm(hidden_size, eps=eps)
        self.dropout = Dropout(hidden_dropout_prob)

    def forward(self, img_feat, img_pos_feat, type_embeddings, img_masks=None):
        """`type_embeddings`: `(type_ids [B, R], token_type table)` — the lookup is then part of the fused sum (what UNITERModelBase passes) —
        or, as in the reference's signature (uniter.py:69-87), a tensor that broadcasts against [B, R, hidden] and is simply added."""
        if isinstance(type_embeddings, torch.Tensor):
            return self._forward_with_type_tensor(img_feat, img_pos_feat, type_embeddings, img_masks)
        type_ids, type_table = type_embeddings
        if F32T.active():      # mmf_amd.fp32_training(): fp32 forward + backward
            if img_masks is not None:
                self.mask_embedding.weight.data[0, :].fill_(0)
            feats = F32T.feature_table_add(img_feat, None if img_masks is None else img_masks.long(), self.mask_embedding.weight, 0)
            transformed_im = self.img_layer_norm(self.img_linear(feats))
            transformed_pos = self.pos_layer_norm(F32T.small_k_linear(img_pos_feat, self.pos_linear.weight, self.pos_linear.bias))
            embeddings = F32T.add_pos_type(F32T.add(transformed_im, transformed_pos), type_ids, None, type_table)
            return self.dropout(self.final_layer_norm(embeddings))
        if F32P.active():      # fp32-accurate forward (mmf_amd.fp32_inference()): same operations on the fp32 kernels
            if img_masks is not None:
                self.mask_embedding.weight.data[0, :].fill_(0)
            feats = F32P.feature_table_add(img_feat, img_masks, self.mask_embedding.weight)
            transformed_im = self.img_layer_norm(self.img_linear(feats))
            transformed_pos = self.pos_layer_norm(F32P.small_k_linear(img_pos_feat, self.pos_linear.weight, self.pos_linear.bias))
            embeddings = F32P.add_pos_type(F32P.add(transformed_im, transformed_pos), type_ids, None, type_table)
            return self.dropout(self.final_layer_norm(embeddings))
        if img_masks is not None:
            self.mask_embedding.weight.data[0, :].fill_(0)                                                   # :76
            feats = Fn.FeatureTableAddFn.apply(img_feat, img_masks.long(), self.mask_embedding.weight, 0)    # :77-78
        else:
            feats = Fn.FeatureTableAddFn.apply(img_feat, None, self.mask_embedding.weight, 0)
        transformed_im = self.img_layer_norm(self.img_linear(feats))                                         # :80
        transformed_pos = self.pos_layer_norm(
            Fn.SmallKLinearFn.apply(img_pos_feat, self.pos_linear.weight, self.pos_linear.bias))             # :81
        embeddings = Fn.AddPosTypeFn.apply(Fn.AddFn.apply(transformed_im, transformed_pos), type_ids, None, type_table)  # :82
        return self.dropout(self.final_layer_norm(embeddings))                                               # :83-84


    def _forward_with_type_tensor(self, img_feat, img_pos_feat, type_embeddings, img_masks):
        if F32T.active() or F32P.active():
            raise NotImplementedError("UNITERImageEmbeddings: a tensor of type embeddings is taken on the bf16 path only; pass (type_ids, table)")
        if img_masks is not None:
            self.mask_embedding.weight.data[0, :].fill_(0)
        feats = Fn.FeatureTableAddFn.apply(img_feat, None if img_masks is None else img_masks.long(), self.mask_embedding.weight, 0)
        transformed_im = self.img_layer_norm(self.img_linear(feats))
        transformed_pos = self.pos_layer_norm(Fn.SmallKLinearFn.apply(img_pos_feat, self.pos_linear.weight, self.pos_linear.bias))
        summed = Fn.AddFn.apply(transformed_im, transformed_pos)
        t = type_embeddings if type_embeddings.is_floating_point() else type_embeddings.float()
        t = t.to(summed.device).expand(summed.shape).contiguous()
        return self.dropout(self.final_layer_norm(Fn.AddFn.apply(summed, t)))


class UNITERModelBase(nn.Module):
    """uniter.py:90-246."""

    def __init__(self, random_init=False, bert_model_name="bert-base-uncased", img_dim=2048, hidden_size=768,
                 hidden_dropout_prob=0, text_embeddings=None, encoder=None):
        super().__init__()
        tcfg = _bert_config(bert_model_name, text_embeddings)
        self.text_embeddings = BertEmbeddingsJit(tcfg)
        self.img_embeddings = UNITERImageEmbeddings(img_dim=img_dim, hidden_size=hidden_size, hidden_dropout_prob=hidden_dropout_prob)
        ecfg = _bert_config(bert_model_name, encoder)
        if ecfg.hidden_size != hidden_size or tcfg.hidden_size != hidden_size:
            raise ValueError("hidden_size (%d) must equal the text-embedding (%d) and encoder (%d) widths" % (
                hidden_size, tcfg.hidden_size, ecfg.hidden_size))
        self.encoder = BertEncoderJit(ecfg)
        self.pooler = BertPooler(ecfg)       # BertModel's pooler: kept for checkpoint compatibility, never called (:150)
        self.apply(lambda m: init_bert_weights(m, ecfg.initializer_range))

    def _compute_txt_embeddings(self, input_ids, position_ids, token_type_ids=None):
        if position_ids is not None:
            T = input_ids.shape[1]
            # (the value check reads the ids back to the host: not possible while a hipGraph is being captured — GraphedTrainStep's eager warm-up
            # passes have run it on the same static batch, and UNITER.add_custom_params builds the ids from the shape alone, uniter.py:738-743)
            capturing = position_ids.is_cuda and torch.cuda.is_current_stream_capturing()
            if position_ids.shape[-1] != T or (not capturing and not bool(
                    (position_ids.reshape(-1, T)[0] == torch.arange(T, device=position_ids.device)).all())):
                raise NotImplementedError("only consecutive position ids 0..T-1 (uniter.py:738-743) are on the fused embedding path")
        return self.text_embeddings(input_ids, token_type_ids)

    def _compute_img_embeddings(self, img_feat, img_pos_feat, img_masks=None, img_type_ids=None):
        if img_type_ids is None:
            img_type_ids = torch.ones(img_feat.shape[:2], dtype=torch.long, device=img_feat.device)          # :172-173
        return self.img_embeddings(img_feat, img_pos_feat, (img_type_ids, self.text_embeddings.token_type_embeddings.weight), img_masks)

    def _compute_img_txt_embeddings(self, input_ids, position_ids, img_feat, img_pos_feat, img_masks=None, txt_type_ids=None,
                                    img_type_ids=None):
        txt_emb = self._compute_txt_embeddings(input_ids, position_ids, txt_type_ids)
        img_emb = self._compute_img_embeddings(img_feat, img_pos_feat, img_masks, img_type_ids)
        if F32T.active():
            return F32T.concat_rows(txt_emb, img_emb)
        return F32P.concat_rows(txt_emb, img_emb) if F32P.active() else Fn.ConcatRowsFn.apply(txt_emb, img_emb)   # :195

    def forward(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, img_masks=None, txt_type_ids=None,
                img_type_ids=None, input_modality="image-text"):
        am = attention_mask.contiguous().long()
        mask_add = torch.empty(am.shape, dtype=torch.float32, device=am.device)
        Fn.nat.make_additive_mask(am, mask_add)                                                              # :210-217
        if input_modality == "image":
            embedding_output = self._compute_img_embeddings(img_feat, img_pos_feat, img_masks, img_type_ids)
        elif input_modality == "text":
            embedding_output = self._compute_txt_embeddings(input_ids, position_ids, txt_type_ids)
        else:
            embedding_output = self._compute_img_txt_embeddings(input_ids, position_ids, img_feat, img_pos_feat, img_masks,
                                                                txt_type_ids, img_type_ids)
        encoded = self.encoder(embedding_output, mask_add.view(am.shape[0], 1, 1, am.shape[1]), output_hidden_states=True)
        return TransformerOutput(encoded[0], encoded[1])          # the reference's named pair (uniter.py:245-247)


def _infer_with_heads(processed_sample_list, uniter_model, heads, losses):
    """uniter.py:249-275."""
    sequence_output = uniter_model(
        processed_sample_list["input_ids"], processed_sample_list["position_ids"], processed_sample_list["image_feat"],
        processed_sample_list["img_pos_feat"], processed_sample_list["attention_mask"],
        img_masks=processed_sample_list["image_mask"])[0]
    dataset_name = processed_sample_list["dataset_name"]
    task = processed_sample_list.get("task", dataset_name)
    outputs = heads[task](sequence_output, processed_sample_list=processed_sample_list)
    if isinstance(outputs, dict) and "losses" in outputs:
        return outputs
    logits = outputs["scores"] if isinstance(outputs, dict) and "scores" in outputs else outputs
    logits = logits.contiguous().view(-1, logits.size(-1))
    output = losses[dataset_name](processed_sample_list, {"scores": logits})
    return {"losses": output, "scores": logits}


DEFAULT_PRETRAINING_HEAD_CONFIGS = {"mlm": {"type": "mlm"}, "itm": {"type": "itm"}, "mrc": {"type": "mrc"}, "mrfr": {"type": "mrfr"},
                                    "wra": {"type": "wra"}}
DEFAULT_PRETRAINING_TASKS = "mlm,itm,mrc,mrfr,wra"


class UNITERForPretraining(nn.Module):
    """uniter.py:350-618.  The per-task preparation is host-side massaging of ids, masks and the feature block (as in the reference);
    encoder and heads are the HIP-backed ones."""

    def __init__(self, head_configs=None, loss_configs=None, tasks=DEFAULT_PRETRAINING_TASKS, mask_probability=0, random_init=False,
                 bert_model_name="bert-base-uncased", img_dim=2048, hidden_size=768, hidden_dropout_prob=0, text_embeddings=None,
                 encoder=None):
        super().__init__()
        if head_configs is None:
            head_configs = copy.deepcopy(DEFAULT_PRETRAINING_HEAD_CONFIGS)
        self.loss_configs = loss_configs if loss_configs is not None else {}
        self.mask_probability = mask_probability
        self.uniter = UNITERModelBase(random_init=random_init, bert_model_name=bert_model_name, img_dim=img_dim,
                                      hidden_size=hidden_size, hidden_dropout_prob=hidden_dropout_prob,
                                      text_embeddings=text_embeddings, encoder=encoder)
        self.heads = nn.ModuleDict()
        self.tasks = tasks.split(",") if isinstance(tasks, str) else list(tasks)
        for task in self.tasks:
            head_config = dict(head_configs[task])
            head_type = head_config.get("type", "mlp")
            head_class = registry.get_transformer_head_class(head_type)
            if head_class is None:
                raise RuntimeError("No transformer head registered for name: %s" % head_type)
            if head_type == "mrfr":
                head_config.pop("type", None)
                self.heads[task] = head_class(self.uniter.img_embeddings.img_linear.weight, **head_config)      # uniter.py:397-400: tied weight
            elif head_type in ("itm", "mlm", "mlp"):
                self.heads[task] = head_class(head_config)                    # uniter.py:400-401
            else:
                head_config.pop("type", None)
                self.heads[task] = head_class(**head_config)                  # :402-403
        self.init_losses()

    def init_losses(self):
        self.losses = nn.ModuleDict()
        for task in self.tasks:
            if task not in self.loss_configs:
                continue          # the head is expected to return a dict with `losses`
            self.losses[task] = MMFLoss(self.loss_configs[task])

    def forward(self, processed_sample_list):
        assert "is_correct" in processed_sample_list, (
            "UNITER pretraining requires mismatched captions. Please add 'false_caption': true under dataset_config in your yaml configs.")
        self._process_sample_list_for_pretraining(processed_sample_list)
        task = processed_sample_list["task"]
        if task == "mlm":
            self._preprocess_mlm(processed_sample_list)
        elif task == "itm":
            self._preprocess_itm(processed_sample_list)
        elif task == "mrc":
            self._preprocess_mrc(processed_sample_list)
        elif task == "mrfr":
            self._preprocess_mrfr(processed_sample_list)
        elif task == "wra":
            self._preprocess_wra(processed_sample_list)
        else:
            raise ValueError("Task %s is not supported for pretraining!" % task)
        return _infer_with_heads(processed_sample_list, self.uniter, self.heads, self.losses)

    # ---- host-side preparation, uniter.py:449-617 ---------------------------------------------------------------------------------------
    def _process_sample_list_for_pretraining(self, processed_sample_list):
        task = processed_sample_list["task"]
        if task in ("mrfr", "mrc"):
            self._add_image_feat_masked(processed_sample_list)
            cls_prob = processed_sample_list["image_info_0"]["cls_prob"]
            processed_sample_list["cls_prob"] = cls_prob if isinstance(cls_prob, torch.Tensor) else torch.tensor(np.array(cls_prob))
        if task not in ("wra", "itm"):
            self._remove_mismatched_captions(processed_sample_list)

    def _add_image_feat_masked(self, processed_sample_list):
        img_feat_masked = torch.clone(processed_sample_list["image_feat"])
        num_feat = img_feat_masked.size(1)
        img_masks = [self._get_img_mask(self.mask_probability, num_feat) for _ in range(img_feat_masked.size(0))]
        img_masks = torch.tensor(img_masks).bool().to(img_feat_masked.device)
        img_masks_ext = img_masks.unsqueeze(-1).expand_as(img_feat_masked)
        processed_sample_list["image_feat_masked"] = img_feat_masked.data.masked_fill(img_masks_ext, 0)
        processed_sample_list["image_mask"] = img_masks

    def _get_img_mask(self, mask_prob, num_bb):
        """uniter.py:480-485, the same generator calls: the seeds of a reference run reproduce its masks."""
        img_mask = list(map(bool, np.random.binomial(1, mask_prob, num_bb)))
        if not any(img_mask):
            img_mask[random.choice(range(num_bb))] = True          # at least one region is masked
        return img_mask

    def _preprocess_mlm(self, processed_sample_list):
        assert "lm_label_ids" in processed_sample_list
        assert "input_ids_masked" in processed_sample_list
        ignore_index = self.heads["mlm"].config.ignore_index
        mlm_labels = {"text": processed_sample_list["lm_label_ids"]}
        mlm_labels["image"] = torch.full(processed_sample_list["image_feat"].shape[:2], fill_value=ignore_index, dtype=torch.long,
                                         device=mlm_labels["text"].device)
        mlm_labels["combined_labels"] = torch.cat([mlm_labels["text"], mlm_labels["image"]], dim=-1)
        processed_sample_list["mlm_labels"] = mlm_labels
        processed_sample_list["input_ids"] = processed_sample_list["input_ids_masked"]

    def _preprocess_itm(self, processed_sample_list):
        assert "is_correct" in processed_sample_list
        processed_sample_list["itm_labels"] = {"is_correct": processed_sample_list["is_correct"]}

    def _get_feature_mask(self, image_mask, sentence_len):
        bs = image_mask.size(0)
        padding_for_txt = torch.zeros((bs, sentence_len)).to(image_mask)
        return torch.cat([padding_for_txt, image_mask], dim=-1)

    def _mask_inputs_in_sample_list(self, processed_sample_list, mask_key):
        assert "image_feat_masked" in processed_sample_list
        sentence_len = processed_sample_list["input_ids"].size(1)
        processed_sample_list[mask_key] = self._get_feature_mask(processed_sample_list["image_mask"], sentence_len)
        processed_sample_list["image_feat"] = processed_sample_list["image_feat_masked"]

    def _preprocess_mrc(self, processed_sample_list):
        assert "cls_prob" in processed_sample_list
        assert "image_mask" in processed_sample_list
        assert "image_feat_masked" in processed_sample_list
        mrc_label_key = self.heads["mrc"].mrc_label_key
        mrc_mask_key = self.heads["mrc"].mrc_mask_key
        image_mask = processed_sample_list["image_mask"]
        cls_prob = processed_sample_list["cls_prob"].to(image_mask.device)
        img_masks_ext = image_mask.unsqueeze(-1).expand_as(cls_prob)
        cls_dim = cls_prob.size(2)
        processed_sample_list[mrc_label_key] = cls_prob[img_masks_ext].contiguous().view(-1, cls_dim)
        self._mask_inputs_in_sample_list(processed_sample_list, mrc_mask_key)

    def _preprocess_mrfr(self, processed_sample_list):
        """uniter.py:542-556: the regression targets are the ORIGINAL features of the masked regions."""
        assert "image_mask" in processed_sample_list
        assert "image_feat_masked" in processed_sample_list
        mrfr_target_key = self.heads["mrfr"].mrfr_target_key
        mrfr_mask_key = self.heads["mrfr"].mrfr_mask_key
        image_mask = processed_sample_list["image_mask"]
        image_feat = processed_sample_list["image_feat"]
        img_masks_ext = image_mask.unsqueeze(-1).expand_as(image_feat)
        feat_dim = image_feat.size(2)
        processed_sample_list[mrfr_target_key] = image_feat[img_masks_ext].contiguous().view(-1, feat_dim)
        self._mask_inputs_in_sample_list(processed_sample_list, mrfr_mask_key)

    def _preprocess_wra(self, processed_sample_list):
        """uniter.py:558-581: padding masks of the optimal-transport alignment from the per-sample lengths (dense batches: no padding)."""
        assert "is_correct" in processed_sample_list
        ot_inputs_key = self.heads["wra"].ot_inputs_key
        wra_label_key = self.heads["wra"].wra_label_key
        txt_lens = [i.size(0) for i in processed_sample_list["input_ids"]]
        num_bbs = [f.size(0) for f in processed_sample_list["image_feat"]]

        def _compute_pad(lens):
            max_len = max(lens)
            pad = torch.zeros(len(lens), max_len)
            for i, n in enumerate(lens):
                pad.data[i, n:].fill_(1)
            return pad

        device = processed_sample_list["input_ids"].device
        processed_sample_list[ot_inputs_key] = {"txt_pad": _compute_pad(txt_lens).to(device).bool(), "img_pad": _compute_pad(num_bbs).to(device).bool()}
        processed_sample_list[wra_label_key] = processed_sample_list["is_correct"]

    def _remove_mismatched_captions(self, processed_sample_list):
        """uniter.py:583-617 selects the matched pairs of each tensor and never writes the selection back: it changes nothing.  Kept
        as the check it effectively is."""
        assert "is_correct" in processed_sample_list


class UNITERForClassification(nn.Module):
    """uniter.py:278-347."""

    def __init__(self, head_configs, loss_configs, tasks, random_init=False, bert_model_name="bert-base-uncased", img_dim=2048,
                 hidden_size=768, hidden_dropout_prob=0, text_embeddings=None, encoder=None):
        super().__init__()
        self.loss_configs = loss_configs
        self.uniter = UNITERModelBase(random_init=random_init, bert_model_name=bert_model_name, img_dim=img_dim,
                                      hidden_size=hidden_size, hidden_dropout_prob=hidden_dropout_prob,
                                      text_embeddings=text_embeddings, encoder=encoder)
        self.heads = nn.ModuleDict()
        self.tasks = tasks.split(",") if isinstance(tasks, str) else list(tasks)
        for task in self.tasks:
            assert task in head_configs, (
                "Task %s is specified in your model configs but there is no head configured for the task. Head configs can be "
                "added under model_config.heads in your yaml configs." % task)
            head_config = head_configs[task]
            head_class = registry.get_transformer_head_class(head_config.get("type", "mlp"))
            if head_class is None:
                raise RuntimeError("No transformer head registered for name: %s" % head_config.get("type", "mlp"))
            self.heads[task] = head_class(head_config)
        self.init_losses()

    def init_losses(self):
        self.losses = nn.ModuleDict()
        for task in self.tasks:
            if task not in self.loss_configs:
                continue
            self.losses[task] = MMFLoss(self.loss_configs[task])

    def forward(self, processed_sample_list):
        return _infer_with_heads(processed_sample_list, self.uniter, self.heads, self.losses)


@registry.register_model("uniter")
class UNITER(BaseModel):
    """uniter.py:621-773."""

    DEFAULTS = dict(random_init=False, bert_model_name="bert-base-uncased", img_dim=2048, hidden_size=768, hidden_dropout_prob=0,
                    text_embeddings={}, encoder={}, losses={}, do_pretraining=False)

    def __init__(self, config):
        super().__init__(config)
        merged = dict(self.DEFAULTS)
        merged.update(dict(config))
        self.config = Config(merged)
        self.do_pretraining = self.config.do_pretraining

    @classmethod
    def config_path(cls):
        return "configs/models/uniter/defaults.yaml"

    def build(self):
        c = self.config
        if self.do_pretraining:
            kw = dict(head_configs=c.get("heads", None), loss_configs=c.losses, random_init=c.random_init, bert_model_name=c.bert_model_name,
                      img_dim=c.img_dim, hidden_size=c.hidden_size, hidden_dropout_prob=c.hidden_dropout_prob,
                      text_embeddings=c.text_embeddings, encoder=c.encoder)
            for key in ("tasks", "mask_probability"):        # uniter.py:658-663: constructor defaults when the key is absent
                if key in c:
                    kw[key] = c[key]
            self.uniter = UNITERForPretraining(**kw)
            tasks = c.get("tasks", DEFAULT_PRETRAINING_TASKS)
            self.tasks = tasks.split(",") if isinstance(tasks, str) else list(tasks)
            return
        self.uniter = UNITERForClassification(
            head_configs=c.heads, loss_configs=c.losses, tasks=c.tasks, random_init=c.random_init, bert_model_name=c.bert_model_name,
            img_dim=c.img_dim, hidden_size=c.hidden_size, hidden_dropout_prob=c.hidden_dropout_prob,
            text_embeddings=c.text_embeddings, encoder=c.encoder)
        self.tasks = self.config.tasks.split(",") if isinstance(self.config.tasks, str) else list(self.config.tasks)

    def init_losses(self):
        """Loss management is deferred to the sub-model (uniter.py:679-684)."""

    def add_pos_feat(self, sample_list):
        """uniter.py:686-717: (x1, y1, x2, y2, w, h, area) from the boxes; a few tiny tensor ops, same test as the reference
        for "boxes still need normalising"."""
        assert "image_info_0" in sample_list
        assert "bbox" in sample_list["image_info_0"]
        feats = sample_list["image_feature_0"]
        info = sample_list["image_info_0"]
        bboxs = torch.as_tensor(info["bbox"], device=feats.device)[:, :, :4].float()
        norm_xy = bboxs.clone()
        # the reference branches on `norm_xy[0, 0, 0] < 1` (uniter.py:697: "boxes still need normalising"), a host read of one device element;
        # the same selection as a device-side `where` keeps the step free of read-backs (capturable as one hipGraph: bench.py GRAPH_CONFIGS)
        if "image_height" in info and "image_width" in info:
            img_h = torch.as_tensor(info["image_height"], device=feats.device).unsqueeze(1).unsqueeze(1)
            img_w = torch.as_tensor(info["image_width"], device=feats.device).unsqueeze(1).unsqueeze(1)
            norm_xy = torch.where(norm_xy[0, 0, 0] < 1, norm_xy / torch.cat([img_w, img_h, img_w, img_h], dim=-1), norm_xy)
        elif bool(norm_xy[0, 0, 0] < 1):     # (no image sizes to divide by: the reference fails with a KeyError here — keep its behaviour)
            raise KeyError("image_height")
        bbox_w = (norm_xy[:, :, 2] - norm_xy[:, :, 0]).unsqueeze(-1)
        bbox_h = (norm_xy[:, :, 3] - norm_xy[:, :, 1]).unsqueeze(-1)
        sample_list["img_pos_feat"] = torch.cat([norm_xy, bbox_w, bbox_h, bbox_w * bbox_h], dim=-1).to(feats)

    def add_custom_params(self, sample_list):
        """uniter.py:719-745."""
        image_feat = sample_list["image_feat"] = sample_list["image_feature_0"]
        image_info = sample_list.get("image_info_0", None) or {}
        image_dim = image_info.get("max_features", None)
        sample_list["image_dim"] = image_dim
        image_mask = torch.arange(image_feat.size(-2), device=image_feat.device).expand(image_feat.size()[:-1])
        if len(image_dim.size()) < len(image_mask.size()):
            image_dim = image_dim.unsqueeze(-1)
            assert len(image_dim.size()) == len(image_mask.size())
        sample_list["image_mask"] = (image_mask < image_dim).long()
        sample_list["attention_mask"] = torch.cat((sample_list["input_mask"], sample_list["image_mask"]), dim=-1)
        task_index = torch.randint(len(self.tasks), (1,)).item()
        sample_list["task"] = self.tasks[task_index]
        sample_list["position_ids"] = torch.arange(0, sample_list["input_ids"].size(1), dtype=torch.long,
                                                   device=image_feat.device).unsqueeze(0)
        self.add_pos_feat(sample_list)
        return sample_list

    def forward(self, sample_list):
        sample_list = self.add_custom_params(sample_list)
        return self.uniter(sample_list)

    def get_optimizer_parameters(self, config):
        from mmf_amd.utils.modeling import get_bert_configured_parameters
        return get_bert_configured_parameters(self)
