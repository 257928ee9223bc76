"""M4C (Iterative Answer Prediction with Pointer-Augmented Multimodal Transformers) behind MMF's model API on the gfx950
kernels (BASELINE.json configs[4]: TextVQA, 20 question + 100 object + 50 OCR + 12 decoding positions).

Mirrors mmf/models/m4c.py: the registered `M4C(BaseModel)` (:34-352) with its `_build_*` / `_forward_*` methods,
`TextBert` (:355-375), `MMT` (:378-458), `OcrPtrNet` (:461-493), `PrevPredEmbeddings` (:496-544), `_get_mask` (:547-553) —
same config keys (`text_bert`, `obj`, `ocr`, `mmt`, `classifier`, `lr_scale_*`, `model_data_dir`, `losses`), the same
registry look-ups (`config.datasets`, `<dataset>_num_final_outputs`, `<dataset>_answer_processor`), the same sample-list
keys and the reference's parameter tree (`text_bert.encoder.layer.*`, `obj_faster_rcnn_fc7.lc.*`,
`linear_ocr_feat_to_mmt_in.*`, `mmt.prev_pred_embeddings.*`, `mmt.encoder.layer.*`, `ocr_ptr_net.query.*`,
`classifier.module.*`), so MMF zoo checkpoints load unmodified.

On the device:
  * the prefix-LM mask the reference materialises as [B, 1, L, L] (:424-440) is a key mask plus a causal-tail length
    handed to the fused attention kernel (`mmf_attn_desc.causal_tail`); the encoder is the one VisualBERT runs on;
  * F.normalize of the appearance / FastText / PHOC features writes bf16 straight into the 3002-wide concatenated OCR
    row, which one MFMA GEMM (zero-padded to 3008 columns) projects;
  * `_batch_gather(cat([ans_emb, ocr_emb]))` is a two-source row gather (no [B, 5050, 768] concatenation);
  * classifier GEMM and pointer-network kernel write the two halves of the [B, T, 5000 + 50] score tensor in place;
  * greedy decoding (:290-305) re-runs only the multimodal transformer per step: `text_bert` is deterministic in eval
    mode, so its output is computed once instead of `dec_step_num` times.

Not built: `text_bert_init_from_bert_base` needs the HF hub (no network) — weights come from a checkpoint or random init;
the `remove_ocr_*` ablation switches zero the corresponding input instead of the normalised feature (same result)."""
import math
import warnings

import torch
from torch import nn

from mmf_amd import functional as Fn
from mmf_amd import fp32_path as F32P
from mmf_amd import fp32_train as F32T
from mmf_amd.common.registry import registry
from mmf_amd.models.base_model import BaseModel
from mmf_amd.modules.encoders import FinetuneFasterRcnnFpnFc7
from mmf_amd.modules.hf_layers import (
    BertConfig, BertEmbeddingsJit, BertEncoderJit, Dropout, LayerNorm, Linear, init_bert_weights)
from mmf_amd.modules.layers import ClassifierLayer
from mmf_amd.utils.configuration import Config, to_container

TEXT_BERT_HIDDEN_SIZE = 768


def _bert_config(overrides):
    d = {k: v for k, v in to_container(overrides or {}).items() if not isinstance(v, (dict, list))}
    return BertConfig.from_dict(d)


def _get_mask(nums, max_num):
    """m4c.py:547-553: b x max_num fp32, 0. on PAD."""
    arange = torch.arange(0, max_num, device=nums.device).unsqueeze(0).expand(nums.size(0), -1)
    return arange.lt(nums.unsqueeze(-1)).type(torch.float32)


def _additive(mask):
    """(1 - mask) * -10000 as fp32 [B, S] (m4c.py:363-364, 439, 475)."""
    m = mask.long().contiguous()
    out = torch.empty(m.shape, dtype=torch.float32, device=m.device)
    Fn.nat.make_additive_mask(m, out)
    return out


class TextBert(nn.Module):
    """m4c.py:355-375 (HF BertEmbeddings + BertEncoder)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = BertEmbeddingsJit(config)
        self.encoder = BertEncoderJit(config)
        self.init_weights()

    def init_weights(self):
        self.apply(lambda m: init_bert_weights(m, self.config.initializer_range))

    def forward(self, txt_inds, txt_mask):
        encoder_inputs = self.embeddings(txt_inds)
        B, T = txt_inds.shape
        extended_attention_mask = _additive(txt_mask).view(B, 1, 1, T)
        return self.encoder(encoder_inputs, extended_attention_mask)[0]


class PrevPredEmbeddings(nn.Module):
    """m4c.py:496-544."""

    def __init__(self, config):
        super().__init__()
        MAX_DEC_LENGTH = 100
        MAX_TYPE_NUM = 5
        hidden_size = config.hidden_size
        ln_eps = config.layer_norm_eps
        self.position_embeddings = nn.Embedding(MAX_DEC_LENGTH, hidden_size)
        self.token_type_embeddings = nn.Embedding(MAX_TYPE_NUM, hidden_size)
        self.ans_layer_norm = LayerNorm(hidden_size, eps=ln_eps)
        self.ocr_layer_norm = LayerNorm(hidden_size, eps=ln_eps)
        self.emb_layer_norm = LayerNorm(hidden_size, eps=ln_eps)
        self.emb_dropout = Dropout(config.hidden_dropout_prob)

    def forward(self, ans_emb, ocr_emb, prev_inds):
        assert prev_inds.dim() == 2 and prev_inds.dtype == torch.long
        assert ans_emb.dim() == 2
        batch_size, seq_length = prev_inds.shape
        ans_num = ans_emb.size(0)
        if F32T.active():      # mmf_amd.fp32_training(): the same operations on the fp32 kernels, with autograd (the classifier weight IS the lookup table: fp32 rows in, fp32 gradient out)
            ans_emb = self.ans_layer_norm(ans_emb)
            ocr_emb = self.ocr_layer_norm(ocr_emb)
            raw_dec_emb = F32T.prev_pred_gather(ans_emb, ocr_emb, prev_inds)
            zero = torch.zeros(batch_size, seq_length, ans_emb.size(-1), dtype=torch.float32, device=ocr_emb.device)
            embeddings = F32T.add_pos_type(zero, prev_inds.ge(ans_num).long(), self.position_embeddings.weight, self.token_type_embeddings.weight)
            return F32T.add(raw_dec_emb, self.emb_dropout(self.emb_layer_norm(embeddings)))
        if F32P.active():      # fp32-accurate forward (mmf_amd.fp32_inference()): the same operations on the fp32 kernels
            ans_emb = self.ans_layer_norm(ans_emb.detach())
            ocr_emb = self.ocr_layer_norm(ocr_emb)
            raw_dec_emb = F32P.prev_pred_gather(ans_emb, ocr_emb, prev_inds)
            zero = torch.zeros(batch_size, seq_length, ans_emb.size(-1), dtype=torch.float32, device=ocr_emb.device)
            embeddings = F32P.add_pos_type(zero, prev_inds.ge(ans_num).long(), self.position_embeddings.weight, self.token_type_embeddings.weight)
            return F32P.add(raw_dec_emb, self.emb_dropout(self.emb_layer_norm(embeddings)))
        if ans_emb.dtype != torch.bfloat16:
            ans_emb = Fn.ParamRowsFn.apply(ans_emb)                     # the classifier weight used as a lookup table (:268)
        ans_emb = self.ans_layer_norm(ans_emb)                          # :523
        ocr_emb = self.ocr_layer_norm(ocr_emb)                          # :524
        assert ans_emb.size(-1) == ocr_emb.size(-1)
        raw_dec_emb = Fn.PrevPredGatherFn.apply(ans_emb, ocr_emb, prev_inds)   # :525-528
        token_type_ids = prev_inds.ge(ans_num).long()                   # :536, 0 -- vocab; 1 -- OCR
        zero = torch.zeros(batch_size, seq_length, ans_emb.size(-1), dtype=torch.bfloat16, device=ocr_emb.device)
        embeddings = Fn.AddPosTypeFn.apply(zero, token_type_ids, self.position_embeddings.weight,
                                           self.token_type_embeddings.weight)   # :531-538
        embeddings = self.emb_dropout(self.emb_layer_norm(embeddings))  # :539-540
        return Fn.AddFn.apply(raw_dec_emb, embeddings)                  # :541


class MMT(nn.Module):
    """m4c.py:378-458."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.prev_pred_embeddings = PrevPredEmbeddings(config)
        self.encoder = BertEncoderJit(config)
        self.init_weights()

    def init_weights(self):
        self.apply(lambda m: init_bert_weights(m, self.config.initializer_range))

    def forward(self, txt_emb, txt_mask, obj_emb, obj_mask, ocr_emb, ocr_mask, fixed_ans_emb, prev_inds):
        dec_emb = self.prev_pred_embeddings(fixed_ans_emb, ocr_emb, prev_inds)                     # :399
        dec_mask = torch.zeros(dec_emb.size(0), dec_emb.size(1), dtype=torch.float32, device=dec_emb.device)   # :405-407
        if F32T.active():
            encoder_inputs = F32T.concat_rows(txt_emb, obj_emb, ocr_emb, dec_emb)
        elif F32P.active():
            encoder_inputs = F32P.concat_rows(txt_emb, obj_emb, ocr_emb, dec_emb)
        else:
            encoder_inputs = Fn.ConcatRowsFn.apply(txt_emb, obj_emb, ocr_emb, dec_emb)             # :408
        attention_mask = torch.cat([txt_mask, obj_mask, ocr_mask, dec_mask], dim=1)                # :409
        txt_max_num, obj_max_num = txt_mask.size(-1), obj_mask.size(-1)
        ocr_max_num, dec_max_num = ocr_mask.size(-1), dec_mask.size(-1)
        # prefix LM (:424-440): all positions see the encoding steps; decoding steps see each other causally
        mask = Fn.PrefixLMMask(_additive(attention_mask), dec_max_num)
        mmt_seq_output = self.encoder(encoder_inputs, mask)[0]
        split = F32T.split_rows if F32T.active() else (F32P.split_rows if F32P.active() else Fn.SplitRowsFn.apply)
        mmt_txt_output, _, mmt_ocr_output, mmt_dec_output = split(
            mmt_seq_output, (txt_max_num, obj_max_num, ocr_max_num, dec_max_num))                  # :446-449
        return {"mmt_seq_output": mmt_seq_output, "mmt_txt_output": mmt_txt_output, "mmt_ocr_output": mmt_ocr_output,
                "mmt_dec_output": mmt_dec_output}


class OcrPtrNet(nn.Module):
    """m4c.py:461-493.  `forward` returns the dynamic OCR scores alone; inside M4C the scores are produced together with
    the fixed-vocabulary scores by `Fn.M4CScoresFn` (one output buffer)."""

    def __init__(self, hidden_size, query_key_size=None):
        super().__init__()
        if query_key_size is None:
            query_key_size = hidden_size
        self.hidden_size = hidden_size
        self.query_key_size = query_key_size
        self.query = Linear(hidden_size, query_key_size)
        self.key = Linear(hidden_size, query_key_size)
        for lin in (self.query, self.key):               # nn.Linear's default init
            nn.init.kaiming_uniform_(lin.weight, a=math.sqrt(5))
            bound = 1.0 / math.sqrt(hidden_size)
            nn.init.uniform_(lin.bias, -bound, bound)

    def forward(self, query_inputs, key_inputs, attention_mask):
        assert attention_mask.dim() == 2
        squeeze_result = query_inputs.dim() == 2
        if squeeze_result:
            query_inputs = query_inputs.unsqueeze(1)
        B, T, _ = query_inputs.shape
        N = key_inputs.shape[1]
        HQ = self.query_key_size
        q = self.query(query_inputs).reshape(B * T, HQ)
        k = self.key(key_inputs).reshape(B * N, HQ)
        out = torch.empty(B * T, N, dtype=torch.float32, device=q.device)
        if q.requires_grad or k.requires_grad:
            raise NotImplementedError("stand-alone OcrPtrNet.forward is inference-only; training goes through Fn.M4CScoresFn")
        Fn.nat.ptr_scores_fwd(q.contiguous(), k.contiguous(), _additive(attention_mask), out, N, B, T, N, HQ, 1.0 / math.sqrt(HQ))
        out = out.view(B, T, N)
        return out.squeeze(1) if squeeze_result else out


@registry.register_model("m4c")
class M4C(BaseModel):
    DEFAULTS = dict(lr_scale_frcn=0.1, lr_scale_text_bert=0.1, lr_scale_mmt=1.0, text_bert_init_from_bert_base=True,
                    text_bert=dict(num_hidden_layers=3), obj=dict(mmt_in_dim=2048, dropout_prob=0.1),
                    ocr=dict(mmt_in_dim=3002, dropout_prob=0.1), mmt=dict(hidden_size=768, num_hidden_layers=4),
                    classifier=dict(type="linear", ocr_max_num=50, ocr_ptr_net=dict(hidden_size=768, query_key_size=768), params={}),
                    model_data_dir="", losses=[dict(type="m4c_decoding_bce_with_mask")])

    def __init__(self, config):
        super().__init__(config)
        merged = to_container(self.DEFAULTS)
        for k, v in to_container(dict(config)).items():
            if isinstance(v, dict) and isinstance(merged.get(k), dict):
                merged[k] = dict(merged[k], **v)
            else:
                merged[k] = v
        self.config = Config(merged)
        self.mmt_config = _bert_config(self.config.mmt)
        self._datasets = registry.get("config").datasets.split(",")

    @classmethod
    def config_path(cls):
        return "configs/models/m4c/defaults.yaml"

    def build(self):
        self.finetune_modules = []           # modules requiring custom learning rates
        self._build_txt_encoding()
        self._build_obj_encoding()
        self._build_ocr_encoding()
        self._build_mmt()
        self._build_output()

    def _build_encoder_config(self):
        return Config({"type": "finetune_faster_rcnn_fpn_fc7",
                       "params": {"in_dim": 2048, "weights_file": "models/detectron.defaults/fc7_w.pkl",
                                  "bias_file": "models/detectron.defaults/fc7_b.pkl", "model_data_dir": self.config.model_data_dir}})

    def _build_fc7(self, key):
        params = dict(self._build_encoder_config().params)
        sub = self.config.get(key, {})
        if "in_dim" in sub:                  # test / benchmark shapes: the reference hard-codes 2048 (:60)
            params["in_dim"] = sub["in_dim"]
        if "fc7_dim" in sub:
            params["out_dim"] = sub["fc7_dim"]
        return FinetuneFasterRcnnFpnFc7(Config(params))

    def _build_txt_encoding(self):
        self.text_bert_config = _bert_config(self.config.text_bert)
        if self.config.text_bert_init_from_bert_base:
            warnings.warn("text_bert_init_from_bert_base: no HF hub in this build; text_bert starts from its own init "
                          "(load an MMF checkpoint to get the BERT-base weights)")
        self.text_bert = TextBert(self.text_bert_config)
        if self.config.text_bert_init_from_bert_base:
            # smaller learning rate on text bert when it starts from BERT_BASE (:78-82)
            self.finetune_modules.append({"module": self.text_bert, "lr_scale": self.config.lr_scale_text_bert})
        # the reference compares against the constant 768 (:88): identical whenever text_bert keeps its default width
        if self.mmt_config.hidden_size != self.text_bert_config.hidden_size:
            self.text_bert_out_linear = Linear(self.text_bert_config.hidden_size, self.mmt_config.hidden_size)
            init_bert_weights(self.text_bert_out_linear, 0.02)
        else:
            self.text_bert_out_linear = nn.Identity()

    def _build_obj_encoding(self):
        H = self.mmt_config.hidden_size
        self.obj_faster_rcnn_fc7 = self._build_fc7("obj")
        self.finetune_modules.append({"module": self.obj_faster_rcnn_fc7, "lr_scale": self.config.lr_scale_frcn})
        self.linear_obj_feat_to_mmt_in = Linear(self.config.obj.mmt_in_dim, H)
        self.linear_obj_bbox_to_mmt_in = Linear(4, H)
        self.obj_feat_layer_norm = LayerNorm(H, eps=1e-5)
        self.obj_bbox_layer_norm = LayerNorm(H, eps=1e-5)
        self.obj_drop = Dropout(self.config.obj.dropout_prob)
        for lin in (self.linear_obj_feat_to_mmt_in, self.linear_obj_bbox_to_mmt_in):
            init_bert_weights(lin, 0.02)

    def _build_ocr_encoding(self):
        H = self.mmt_config.hidden_size
        ocr = self.config.ocr
        self.remove_ocr_fasttext = ocr.get("remove_ocr_fasttext", False)
        self.remove_ocr_phoc = ocr.get("remove_ocr_phoc", False)
        self.remove_ocr_frcn = ocr.get("remove_ocr_frcn", False)
        self.remove_ocr_semantics = ocr.get("remove_ocr_semantics", False)
        self.remove_ocr_bbox = ocr.get("remove_ocr_bbox", False)
        self.ocr_faster_rcnn_fc7 = self._build_fc7("ocr")
        self.finetune_modules.append({"module": self.ocr_faster_rcnn_fc7, "lr_scale": self.config.lr_scale_frcn})
        self.linear_ocr_feat_to_mmt_in = Linear(ocr.mmt_in_dim, H)
        self.linear_ocr_bbox_to_mmt_in = Linear(4, H)
        self.ocr_feat_layer_norm = LayerNorm(H, eps=1e-5)
        self.ocr_bbox_layer_norm = LayerNorm(H, eps=1e-5)
        self.ocr_drop = Dropout(ocr.dropout_prob)
        for lin in (self.linear_ocr_feat_to_mmt_in, self.linear_ocr_bbox_to_mmt_in):
            init_bert_weights(lin, 0.02)

    def _build_mmt(self):
        self.mmt = MMT(self.mmt_config)
        self.finetune_modules.append({"module": self.mmt, "lr_scale": self.config.lr_scale_mmt})

    def _build_output(self):
        self.ocr_ptr_net = OcrPtrNet(**to_container(self.config.classifier.ocr_ptr_net))
        num_choices = registry.get(self._datasets[0] + "_num_final_outputs")
        num_choices -= self.config.classifier.ocr_max_num      # OCR copying is scored by the pointer network (:159-163)
        self.classifier = ClassifierLayer(self.config.classifier.type, in_dim=self.mmt_config.hidden_size, out_dim=num_choices,
                                          **to_container(self.config.classifier.params))
        self.answer_processor = registry.get(self._datasets[0] + "_answer_processor")

    # ------------------------------------------------------------------------------------------
    def forward(self, sample_list):
        fwd_results = {}
        self._forward_txt_encoding(sample_list, fwd_results)
        self._forward_obj_encoding(sample_list, fwd_results)
        self._forward_ocr_encoding(sample_list, fwd_results)
        self._forward_mmt_and_output(sample_list, fwd_results)
        return {"scores": fwd_results["scores"]}

    def _forward_txt_encoding(self, sample_list, fwd_results):
        fwd_results["txt_inds"] = sample_list["text"]
        fwd_results["txt_mask"] = _get_mask(sample_list["text_len"], sample_list["text"].size(1))

    def _forward_obj_encoding(self, sample_list, fwd_results):
        obj_fc6 = sample_list["image_feature_0"]
        # fp32: mmf_amd.fp32_training() (F32T: with autograd) or the fp32-accurate forward mmf_amd.fp32_inference() (F32P): the same operations on the fp32 kernels
        X = F32T if F32T.active() else (F32P if F32P.active() else None)
        obj_fc7 = (X.l2norm_rows if X else Fn.L2NormRowsFn.apply)(self.obj_faster_rcnn_fc7(obj_fc6))   # :193-195
        feat = self.obj_feat_layer_norm(self.linear_obj_feat_to_mmt_in(obj_fc7))
        bbox = self.obj_bbox_layer_norm((X.small_k_linear if X else Fn.SmallKLinearFn.apply)(
            sample_list["obj_bbox_coordinates"], self.linear_obj_bbox_to_mmt_in.weight, self.linear_obj_bbox_to_mmt_in.bias))
        fwd_results["obj_mmt_in"] = self.obj_drop((X.add if X else Fn.AddFn.apply)(feat, bbox))        # :199-203
        obj_nums = sample_list["image_info_0"]["max_features"]
        fwd_results["obj_mask"] = _get_mask(obj_nums, obj_fc6.size(1))

    def _forward_ocr_encoding(self, sample_list, fwd_results):
        ocr_fasttext = sample_list["context_feature_0"]
        assert ocr_fasttext.size(-1) == 300
        ocr_phoc = sample_list["context_feature_1"]
        assert ocr_phoc.size(-1) == 604
        ocr_fc6 = sample_list["image_feature_1"][:, : ocr_fasttext.size(1), :]
        order_dim = sample_list["order_vectors"].size(-1)          # legacy LoRRA order vectors: all zeros (:225-227)
        if self.remove_ocr_fasttext or self.remove_ocr_semantics:
            ocr_fasttext = torch.zeros_like(ocr_fasttext)
        if self.remove_ocr_phoc or self.remove_ocr_semantics:
            ocr_phoc = torch.zeros_like(ocr_phoc)
        if self.remove_ocr_frcn or self.remove_ocr_semantics:
            ocr_fc6 = torch.zeros_like(ocr_fc6)
        ocr_fc7 = self.ocr_faster_rcnn_fc7(ocr_fc6.contiguous())
        if self.remove_ocr_frcn or self.remove_ocr_semantics:
            ocr_fc7 = ocr_fc7.detach() * 0
        ocr_bbox = sample_list["ocr_bbox_coordinates"]
        if self.remove_ocr_bbox:
            ocr_bbox = torch.zeros_like(ocr_bbox)
        if F32T.active():
            ocr_feat = F32T.ocr_feature_concat(ocr_fasttext, ocr_phoc, ocr_fc7, order_dim)
            feat = self.ocr_feat_layer_norm(F32T.padded_linear(ocr_feat, self.linear_ocr_feat_to_mmt_in.weight, self.linear_ocr_feat_to_mmt_in.bias))
            bbox = self.ocr_bbox_layer_norm(F32T.small_k_linear(ocr_bbox, self.linear_ocr_bbox_to_mmt_in.weight, self.linear_ocr_bbox_to_mmt_in.bias))
            fwd_results["ocr_mmt_in"] = self.ocr_drop(F32T.add(feat, bbox))
        elif F32P.active():
            ocr_feat, _ = F32P.ocr_feature_concat(ocr_fasttext, ocr_phoc, ocr_fc7, order_dim)
            feat = self.ocr_feat_layer_norm(F32P.padded_linear(ocr_feat, self.linear_ocr_feat_to_mmt_in.weight, self.linear_ocr_feat_to_mmt_in.bias))
            bbox = self.ocr_bbox_layer_norm(F32P.small_k_linear(ocr_bbox, self.linear_ocr_bbox_to_mmt_in.weight, self.linear_ocr_bbox_to_mmt_in.bias))
            fwd_results["ocr_mmt_in"] = self.ocr_drop(F32P.add(feat, bbox))
        else:
            ocr_feat = Fn.OcrFeatureConcatFn.apply(ocr_fasttext, ocr_phoc, ocr_fc7, order_dim)              # :211-237
            feat = self.ocr_feat_layer_norm(Fn.PaddedLinearFn.apply(
                ocr_feat, self.linear_ocr_feat_to_mmt_in.weight, self.linear_ocr_feat_to_mmt_in.bias))
            bbox = self.ocr_bbox_layer_norm(Fn.SmallKLinearFn.apply(
                ocr_bbox, self.linear_ocr_bbox_to_mmt_in.weight, self.linear_ocr_bbox_to_mmt_in.bias))
            fwd_results["ocr_mmt_in"] = self.ocr_drop(Fn.AddFn.apply(feat, bbox))                            # :243-247
        ocr_nums = sample_list["context_info_0"]["max_features"]
        fwd_results["ocr_mask"] = _get_mask(ocr_nums, ocr_fasttext.size(1))

    def _forward_mmt(self, sample_list, fwd_results):
        if "txt_emb" not in fwd_results or self.training:
            text_bert_out = self.text_bert(txt_inds=fwd_results["txt_inds"], txt_mask=fwd_results["txt_mask"])
            fwd_results["txt_emb"] = self.text_bert_out_linear(text_bert_out)
        mmt_results = self.mmt(
            txt_emb=fwd_results["txt_emb"], txt_mask=fwd_results["txt_mask"], obj_emb=fwd_results["obj_mmt_in"],
            obj_mask=fwd_results["obj_mask"], ocr_emb=fwd_results["ocr_mmt_in"], ocr_mask=fwd_results["ocr_mask"],
            fixed_ans_emb=self.classifier.module.weight, prev_inds=fwd_results["prev_inds"])
        fwd_results.update(mmt_results)

    def _forward_output(self, sample_list, fwd_results):
        cls, ptr = self.classifier.module, self.ocr_ptr_net
        if F32T.active():
            fwd_results["scores"] = F32T.m4c_scores(fwd_results["mmt_dec_output"], fwd_results["mmt_ocr_output"], cls.weight, cls.bias, ptr.query.weight,
                                                    ptr.query.bias, ptr.key.weight, ptr.key.bias, _additive(fwd_results["ocr_mask"]))
            return
        if F32P.active():
            fwd_results["scores"] = F32P.m4c_scores(fwd_results["mmt_dec_output"], fwd_results["mmt_ocr_output"], cls.weight, cls.bias, ptr.query.weight,
                                                    ptr.query.bias, ptr.key.weight, ptr.key.bias, _additive(fwd_results["ocr_mask"]))
            return
        fwd_results["scores"] = Fn.M4CScoresFn.apply(
            fwd_results["mmt_dec_output"], fwd_results["mmt_ocr_output"], cls.weight, cls.bias, ptr.query.weight, ptr.query.bias,
            ptr.key.weight, ptr.key.bias, _additive(fwd_results["ocr_mask"]), Fn.shadows.get(cls.weight),
            Fn.shadows.get(ptr.query.weight), Fn.shadows.get(ptr.key.weight))                                # :280-283

    def _forward_mmt_and_output(self, sample_list, fwd_results):
        if self.training:
            fwd_results["prev_inds"] = sample_list["train_prev_inds"].clone()
            self._forward_mmt(sample_list, fwd_results)
            self._forward_output(sample_list, fwd_results)
        elif self.config.get("kv_cached_decode", True) and not F32P.active() and not F32T.active():
            self._decode_incremental(sample_list, fwd_results)
        else:
            # the reference's loop, kept for A/B tests: the whole multimodal transformer once per decoding step (:290-305)
            dec_step_num = sample_list["train_prev_inds"].size(1)
            # fill prev_inds with BOS_IDX at index 0, and zeros elsewhere
            fwd_results["prev_inds"] = torch.zeros_like(sample_list["train_prev_inds"])
            fwd_results["prev_inds"][:, 0] = self.answer_processor.BOS_IDX
            # greedy decoding at test time
            for _ in range(dec_step_num):
                self._forward_mmt(sample_list, fwd_results)
                self._forward_output(sample_list, fwd_results)
                argmax_inds = fwd_results["scores"].argmax(dim=-1)
                fwd_results["prev_inds"][:, 1:] = argmax_inds[:, :-1]

    @torch.no_grad()
    def _decode_incremental(self, sample_list, fwd_results):
        """Greedy decoding (m4c.py:290-305) WITHOUT re-encoding: the reference re-runs the multimodal transformer over all
        T + O + N + D positions D times; under its prefix-LM mask (:424-440) the T + O + N encoder positions never see a
        decoding position, so they — and their keys / values in every layer — are computed ONCE, and decoding step i runs a
        single new row per sample through the layers against the per-layer K|V cache (mmf_amd/modules/infer.py).  Row i of
        the scores is final the moment it is produced (it only ever depends on predictions < i), so the result equals the
        last iteration of the reference's loop."""
        from mmf_amd.modules.infer import layer_rows, new_cache
        if "txt_emb" not in fwd_results:
            text_bert_out = self.text_bert(txt_inds=fwd_results["txt_inds"], txt_mask=fwd_results["txt_mask"])
            fwd_results["txt_emb"] = self.text_bert_out_linear(text_bert_out)
        txt_emb, obj_emb, ocr_emb = fwd_results["txt_emb"], fwd_results["obj_mmt_in"], fwd_results["ocr_mmt_in"]
        txt_mask, obj_mask, ocr_mask = fwd_results["txt_mask"], fwd_results["obj_mask"], fwd_results["ocr_mask"]
        B, T, H = txt_emb.shape
        O, N = obj_emb.shape[1], ocr_emb.shape[1]
        D = sample_list["train_prev_inds"].size(1)
        E = T + O + N
        dev = txt_emb.device
        layers = self.mmt.encoder.layer
        # keys: the encoder positions by their masks, the decoding positions all visible (step i only addresses keys <= E + i)
        key_mask = _additive(torch.cat([txt_mask, obj_mask, ocr_mask, torch.ones(B, D, dtype=txt_mask.dtype, device=dev)], dim=1))
        cache = new_cache(B, E + D, H, len(layers), dev)
        x = Fn.ConcatRowsFn.apply(txt_emb, obj_emb, ocr_emb).reshape(B * E, H)
        for l, layer in enumerate(layers):
            x = layer_rows(layer, x, cache[l], B, E, E + D, 0, E, key_mask)
        enc_out = x.view(B, E, H)
        mmt_ocr_output = enc_out[:, T + O:T + O + N].contiguous()
        pp = self.mmt.prev_pred_embeddings
        cls, ptr = self.classifier.module, self.ocr_ptr_net
        V = cls.weight.shape[0]
        ans_emb = pp.ans_layer_norm(Fn.ParamRowsFn.apply(cls.weight))          # PrevPredEmbeddings.forward, :523-524, once
        ocr_ln = pp.ocr_layer_norm(ocr_emb)
        prev = torch.zeros_like(sample_list["train_prev_inds"])
        prev[:, 0] = self.answer_processor.BOS_IDX
        scores = torch.empty(B, D, V + N, dtype=torch.float32, device=dev)
        zero = torch.zeros(B, 1, H, dtype=torch.bfloat16, device=dev)
        ocr_mask_add = _additive(ocr_mask)
        w_cls16, w_q16, w_k16 = Fn.shadows.get(cls.weight), Fn.shadows.get(ptr.query.weight), Fn.shadows.get(ptr.key.weight)
        for i in range(D):
            inds = prev[:, i:i + 1].contiguous()
            raw = Fn.PrevPredGatherFn.apply(ans_emb, ocr_ln, inds)                                      # :525-528
            emb = Fn.AddPosTypeFn.apply(zero, inds.ge(V).long(), pp.position_embeddings.weight[i:i + 1],
                                        pp.token_type_embeddings.weight)                                  # :531-538, position i
            x = Fn.AddFn.apply(raw, pp.emb_layer_norm(emb)).reshape(B, H)                                # :539-541 (dropout off)
            for l, layer in enumerate(layers):
                x = layer_rows(layer, x, cache[l], B, 1, E + D, E + i, E + i + 1, key_mask)
            s_i = Fn.M4CScoresFn.apply(x.view(B, 1, H), mmt_ocr_output, cls.weight, cls.bias, ptr.query.weight, ptr.query.bias,
                                       ptr.key.weight, ptr.key.bias, ocr_mask_add, w_cls16, w_q16, w_k16)    # :275-283
            scores[:, i] = s_i[:, 0]
            if i + 1 < D:
                prev[:, i + 1] = s_i[:, 0].argmax(dim=-1)
        fwd_results["scores"] = scores
        fwd_results["prev_inds"] = prev
        fwd_results["mmt_ocr_output"] = mmt_ocr_output

    def get_optimizer_parameters(self, config):
        """m4c.py:307-329."""
        optimizer_param_groups = []
        base_lr = config.optimizer.params.lr
        finetune_params_set = set()
        for m in self.finetune_modules:
            optimizer_param_groups.append({"params": list(m["module"].parameters()), "lr": base_lr * m["lr_scale"]})
            finetune_params_set.update(list(m["module"].parameters()))
        remaining_params = [p for p in self.parameters() if p not in finetune_params_set]
        optimizer_param_groups.insert(0, {"params": remaining_params})
        return optimizer_param_groups

    @classmethod
    def update_registry_for_pretrained(cls, config, checkpoint, full_output):
        """m4c.py:331-352."""
        datasets = full_output["full_config"].datasets
        dataset = datasets.split(",")[0]
        registry.register("config", Config({"datasets": datasets}))
        registry.register("%s_num_final_outputs" % dataset,
                          checkpoint["classifier.module.weight"].size(0) + config.classifier.ocr_max_num)
        registry.register("%s_answer_processor" % dataset, Config({"BOS_IDX": 1}))
