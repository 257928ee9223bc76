"""CPU oracle for MMF's M4C path (BASELINE.json configs[4]) — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

fp32 PyTorch restatement of mmf/models/m4c.py: `M4C._forward_txt_encoding` (:183-189), `_forward_obj_encoding` (:191-207),
`_forward_ocr_encoding` (:209-253), `_forward_mmt` (:255-273), `_forward_output` (:275-283), the train / greedy-decode switch
`_forward_mmt_and_output` (:285-305), `TextBert.forward` (:359-375), `MMT.forward` (:386-458, the prefix-LM mask at :424-440),
`OcrPtrNet.forward` (:474-493), `PrevPredEmbeddings.forward` (:513-544), `_get_mask` (:547-553), `_batch_gather` (:566-578) and
`FinetuneFasterRcnnFpnFc7.forward` (mmf/modules/encoders.py:177-180), over the encoder layer restated in
oracle/visual_bert_oracle.py (`bert_layer`); loss `M4CDecodingBCEWithMaskLoss.forward` (mmf/modules/losses.py:581-592).

Parity status: PINNED against `tests/golden/m4c_small64.npz`, produced by running the reference's own `M4C._forward_*`
methods, `TextBert`, `MMT`, `OcrPtrNet`, `PrevPredEmbeddings` and the loss (tests/golden/make_golden.py, `make_m4c`).
Parameter names are the reference's (`text_bert.*`, `obj_faster_rcnn_fc7.lc.*`, `linear_obj_feat_to_mmt_in.*`, `mmt.*`,
`ocr_ptr_net.*`, `classifier.module.*`).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from oracle.visual_bert_oracle import bert_layer, layer_norm

DEFAULT_CONFIG = dict(
    # text_bert (m4c/defaults.yaml text_bert + bert-base-uncased)
    text_hidden_size=768, text_num_hidden_layers=3, text_num_attention_heads=12, text_intermediate_size=3072,
    vocab_size=30522, max_position_embeddings=512, type_vocab_size=2,
    # mmt
    hidden_size=768, num_hidden_layers=4, num_attention_heads=12, intermediate_size=3072,
    layer_norm_eps=1e-12, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
    obj_in_dim=2048, obj_fc7_dim=2048, ocr_in_dim=2048, ocr_fc7_dim=2048, fasttext_dim=300, phoc_dim=604, ocr_max_num=50,
    obj_dropout_prob=0.1, ocr_dropout_prob=0.1, num_choices=5000, query_key_size=768, max_dec_length=100, max_type_num=5,
    bos_idx=1, pad_token_id=0,
)


def _encoder_shapes(s, prefix, H, I, L):
    for i in range(L):
        p = prefix + "encoder.layer.%d." % i
        for n in ("query", "key", "value"):
            s[p + "attention.self.%s.weight" % n] = (H, H)
            s[p + "attention.self.%s.bias" % n] = (H,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "attention.output.LayerNorm.weight"] = (H,)
        s[p + "attention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (I, H)
        s[p + "intermediate.dense.bias"] = (I,)
        s[p + "output.dense.weight"] = (H, I)
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)


def parameter_shapes(cfg):
    """State dict of the reference M4C (m4c.py:46-170), in module-registration order."""
    H, TH = cfg["hidden_size"], cfg["text_hidden_size"]
    s = OrderedDict()
    e = "text_bert.embeddings."
    s[e + "word_embeddings.weight"] = (cfg["vocab_size"], TH)
    s[e + "position_embeddings.weight"] = (cfg["max_position_embeddings"], TH)
    s[e + "token_type_embeddings.weight"] = (cfg["type_vocab_size"], TH)
    s[e + "LayerNorm.weight"] = (TH,)
    s[e + "LayerNorm.bias"] = (TH,)
    _encoder_shapes(s, "text_bert.", TH, cfg["text_intermediate_size"], cfg["text_num_hidden_layers"])
    if TH != H:
        s["text_bert_out_linear.weight"] = (H, TH)
        s["text_bert_out_linear.bias"] = (H,)
    s["obj_faster_rcnn_fc7.lc.weight"] = (cfg["obj_fc7_dim"], cfg["obj_in_dim"])
    s["obj_faster_rcnn_fc7.lc.bias"] = (cfg["obj_fc7_dim"],)
    s["linear_obj_feat_to_mmt_in.weight"] = (H, cfg["obj_fc7_dim"])
    s["linear_obj_feat_to_mmt_in.bias"] = (H,)
    s["linear_obj_bbox_to_mmt_in.weight"] = (H, 4)
    s["linear_obj_bbox_to_mmt_in.bias"] = (H,)
    for n in ("obj_feat_layer_norm", "obj_bbox_layer_norm"):
        s[n + ".weight"] = (H,)
        s[n + ".bias"] = (H,)
    s["ocr_faster_rcnn_fc7.lc.weight"] = (cfg["ocr_fc7_dim"], cfg["ocr_in_dim"])
    s["ocr_faster_rcnn_fc7.lc.bias"] = (cfg["ocr_fc7_dim"],)
    ocr_in = cfg["fasttext_dim"] + cfg["phoc_dim"] + cfg["ocr_fc7_dim"] + cfg["ocr_max_num"]
    s["linear_ocr_feat_to_mmt_in.weight"] = (H, ocr_in)
    s["linear_ocr_feat_to_mmt_in.bias"] = (H,)
    s["linear_ocr_bbox_to_mmt_in.weight"] = (H, 4)
    s["linear_ocr_bbox_to_mmt_in.bias"] = (H,)
    for n in ("ocr_feat_layer_norm", "ocr_bbox_layer_norm"):
        s[n + ".weight"] = (H,)
        s[n + ".bias"] = (H,)
    p = "mmt.prev_pred_embeddings."
    s[p + "position_embeddings.weight"] = (cfg["max_dec_length"], H)
    s[p + "token_type_embeddings.weight"] = (cfg["max_type_num"], H)
    for n in ("ans_layer_norm", "ocr_layer_norm", "emb_layer_norm"):
        s[p + n + ".weight"] = (H,)
        s[p + n + ".bias"] = (H,)
    _encoder_shapes(s, "mmt.", H, cfg["intermediate_size"], cfg["num_hidden_layers"])
    s["ocr_ptr_net.query.weight"] = (cfg["query_key_size"], H)
    s["ocr_ptr_net.query.bias"] = (cfg["query_key_size"],)
    s["ocr_ptr_net.key.weight"] = (cfg["query_key_size"], H)
    s["ocr_ptr_net.key.bias"] = (cfg["query_key_size"],)
    s["classifier.module.weight"] = (cfg["num_choices"], H)
    s["classifier.module.bias"] = (cfg["num_choices"],)
    return s


def _stack(sd, prefix):
    """The encoder restatement in visual_bert_oracle indexes `bert.encoder.layer.N.*`."""
    return {k.replace(prefix + "encoder.", "bert.encoder."): v for k, v in sd.items() if k.startswith(prefix + "encoder.")}


def _stack_cfg(cfg, text):
    pre = "text_" if text else ""
    return dict(hidden_size=cfg[pre + "hidden_size"], num_attention_heads=cfg[pre + "num_attention_heads"],
                layer_norm_eps=cfg["layer_norm_eps"])


def get_mask(nums, max_num):
    """_get_mask, m4c.py:547-553: fp32, 0 on PAD."""
    ar = torch.arange(0, max_num, device=nums.device).unsqueeze(0).expand(nums.size(0), -1)
    return ar.lt(nums.unsqueeze(-1)).float()


def fc7(sd, prefix, x, gate=None):
    """FinetuneFasterRcnnFpnFc7.forward, encoders.py:177-180.  `gate` (bool, shape of the output) replaces the ReLU's own
    sign decision: parity tests of a reduced-precision implementation pass ITS gates so that pre-activations within
    rounding noise of zero do not turn into O(1) gradient differences (same device as `pooler_masks` in vilbert_oracle)."""
    pre = F.linear(x, sd[prefix + "lc.weight"], sd[prefix + "lc.bias"])
    return F.relu(pre) if gate is None else pre * gate.to(pre.dtype)


def text_bert(sd, cfg, txt_inds, txt_mask, train=False):
    """TextBert.forward, m4c.py:359-375 (HF BertEmbeddings: token types 0, positions arange)."""
    hd = cfg["hidden_dropout_prob"] if train else 0.0
    ad = cfg["attention_probs_dropout_prob"] if train else 0.0
    e = "text_bert.embeddings."
    T = txt_inds.size(1)
    pos = torch.arange(T, device=txt_inds.device).unsqueeze(0).expand(txt_inds.shape)
    emb = (F.embedding(txt_inds, sd[e + "word_embeddings.weight"], padding_idx=cfg.get("pad_token_id", 0))
           + F.embedding(pos, sd[e + "position_embeddings.weight"])
           + F.embedding(torch.zeros_like(txt_inds), sd[e + "token_type_embeddings.weight"]))
    h = layer_norm(emb, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], cfg["layer_norm_eps"])
    h = F.dropout(h, hd, training=hd > 0)
    ext = (1.0 - txt_mask[:, None, None, :]) * -10000.0  # :363-364
    st, sc = _stack(sd, "text_bert."), _stack_cfg(cfg, True)
    for i in range(cfg["text_num_hidden_layers"]):
        h, _ = bert_layer(st, sc, i, h, ext, hd, ad)
    return h


def obj_encoding(sd, cfg, sample_list, train=False, gate=None):
    """M4C._forward_obj_encoding, m4c.py:191-207.  Returns (obj_mmt_in, obj_mask)."""
    obj_fc7 = F.normalize(fc7(sd, "obj_faster_rcnn_fc7.", sample_list["image_feature_0"], gate), dim=-1)  # :193-195
    x = (F.layer_norm(F.linear(obj_fc7, sd["linear_obj_feat_to_mmt_in.weight"], sd["linear_obj_feat_to_mmt_in.bias"]),
                      (cfg["hidden_size"],), sd["obj_feat_layer_norm.weight"], sd["obj_feat_layer_norm.bias"], 1e-5)
         + F.layer_norm(F.linear(sample_list["obj_bbox_coordinates"], sd["linear_obj_bbox_to_mmt_in.weight"],
                                 sd["linear_obj_bbox_to_mmt_in.bias"]),
                        (cfg["hidden_size"],), sd["obj_bbox_layer_norm.weight"], sd["obj_bbox_layer_norm.bias"], 1e-5))  # :199-201
    p = cfg["obj_dropout_prob"] if train else 0.0
    x = F.dropout(x, p, training=p > 0)  # :202
    return x, get_mask(sample_list["image_info_0"]["max_features"], x.size(1))  # :206-207


def ocr_encoding(sd, cfg, sample_list, train=False, gate=None):
    """M4C._forward_ocr_encoding, m4c.py:209-253 (no `remove_ocr_*` ablations).  Returns (ocr_mmt_in, ocr_mask)."""
    ft = F.normalize(sample_list["context_feature_0"], dim=-1)  # :211-212
    assert ft.size(-1) == 300
    ph = F.normalize(sample_list["context_feature_1"], dim=-1)  # :216-217
    assert ph.size(-1) == 604
    ocr_fc6 = sample_list["image_feature_1"][:, : ft.size(1), :]  # :221
    f7 = F.normalize(fc7(sd, "ocr_faster_rcnn_fc7.", ocr_fc6, gate), dim=-1)  # :222-223
    order = torch.zeros_like(sample_list["order_vectors"])  # :227
    feat = torch.cat([ft, ph, f7, order], dim=-1)  # :235-237
    x = (F.layer_norm(F.linear(feat, sd["linear_ocr_feat_to_mmt_in.weight"], sd["linear_ocr_feat_to_mmt_in.bias"]),
                      (cfg["hidden_size"],), sd["ocr_feat_layer_norm.weight"], sd["ocr_feat_layer_norm.bias"], 1e-5)
         + F.layer_norm(F.linear(sample_list["ocr_bbox_coordinates"], sd["linear_ocr_bbox_to_mmt_in.weight"],
                                 sd["linear_ocr_bbox_to_mmt_in.bias"]),
                        (cfg["hidden_size"],), sd["ocr_bbox_layer_norm.weight"], sd["ocr_bbox_layer_norm.bias"], 1e-5))  # :243-245
    p = cfg["ocr_dropout_prob"] if train else 0.0
    x = F.dropout(x, p, training=p > 0)  # :246
    return x, get_mask(sample_list["context_info_0"]["max_features"], x.size(1))  # :250-251


def prev_pred_embeddings(sd, cfg, ans_emb, ocr_emb, prev_inds, train=False):
    """PrevPredEmbeddings.forward, m4c.py:513-544."""
    p = "mmt.prev_pred_embeddings."
    eps = cfg["layer_norm_eps"]
    B, T = prev_inds.shape
    ans_num = ans_emb.size(0)
    ans = layer_norm(ans_emb, sd[p + "ans_layer_norm.weight"], sd[p + "ans_layer_norm.bias"], eps)  # :523
    ocr = layer_norm(ocr_emb, sd[p + "ocr_layer_norm.weight"], sd[p + "ocr_layer_norm.bias"], eps)  # :524
    cat = torch.cat([ans.unsqueeze(0).expand(B, -1, -1), ocr], dim=1)  # :526-527
    flat = cat.reshape(B * cat.size(1), cat.size(2))
    raw = F.embedding(torch.arange(B, device=prev_inds.device).unsqueeze(-1) * cat.size(1) + prev_inds, flat)  # :528, _batch_gather
    pos = torch.arange(T, device=prev_inds.device).unsqueeze(0).expand(B, T)
    typ = prev_inds.ge(ans_num).long()  # :536
    emb = F.embedding(pos, sd[p + "position_embeddings.weight"]) + F.embedding(typ, sd[p + "token_type_embeddings.weight"])
    emb = layer_norm(emb, sd[p + "emb_layer_norm.weight"], sd[p + "emb_layer_norm.bias"], eps)  # :539
    hd = cfg["hidden_dropout_prob"] if train else 0.0
    emb = F.dropout(emb, hd, training=hd > 0)  # :540
    return raw + emb  # :541


def prefix_lm_mask(attention_mask, dec_num):
    """The [B, 1, L, L] additive mask of MMT.forward, m4c.py:424-440: every position sees the (non-padded) encoding
    positions; decoding positions additionally see themselves and earlier decoding positions; nothing else sees them."""
    L = attention_mask.size(1)
    ext = attention_mask[:, None, None, :].repeat(1, 1, L, 1)
    ext[:, :, -dec_num:, -dec_num:] = torch.tril(torch.ones(dec_num, dec_num, device=attention_mask.device))
    return (1.0 - ext) * -10000.0


def mmt(sd, cfg, txt_emb, txt_mask, obj_emb, obj_mask, ocr_emb, ocr_mask, fixed_ans_emb, prev_inds, train=False):
    """MMT.forward, m4c.py:386-458."""
    hd = cfg["hidden_dropout_prob"] if train else 0.0
    ad = cfg["attention_probs_dropout_prob"] if train else 0.0
    dec_emb = prev_pred_embeddings(sd, cfg, fixed_ans_emb, ocr_emb, prev_inds, train)  # :399
    dec_mask = torch.zeros(dec_emb.size(0), dec_emb.size(1), dtype=torch.float32, device=dec_emb.device)  # :405-407
    h = torch.cat([txt_emb, obj_emb, ocr_emb, dec_emb], dim=1)  # :408
    am = torch.cat([txt_mask, obj_mask, ocr_mask, dec_mask], dim=1)  # :409
    T, O, N, D = txt_mask.size(-1), obj_mask.size(-1), ocr_mask.size(-1), dec_mask.size(-1)
    ext = prefix_lm_mask(am, D)
    st, sc = _stack(sd, "mmt."), _stack_cfg(cfg, False)
    for i in range(cfg["num_hidden_layers"]):
        h, _ = bert_layer(st, sc, i, h, ext, hd, ad)
    return {"mmt_seq_output": h, "mmt_txt_output": h[:, :T], "mmt_ocr_output": h[:, T + O:T + O + N],
            "mmt_dec_output": h[:, -D:]}  # :446-458


def ocr_ptr_net(sd, cfg, query_inputs, key_inputs, attention_mask):
    """OcrPtrNet.forward, m4c.py:474-493."""
    ext = ((1.0 - attention_mask) * -10000.0).unsqueeze(1)
    q = F.linear(query_inputs, sd["ocr_ptr_net.query.weight"], sd["ocr_ptr_net.query.bias"])
    k = F.linear(key_inputs, sd["ocr_ptr_net.key.weight"], sd["ocr_ptr_net.key.bias"])
    return torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(cfg["query_key_size"]) + ext


def forward_mmt_and_output(sd, cfg, sample_list, enc, prev_inds, train=False):
    """M4C._forward_mmt + _forward_output, m4c.py:255-283, for one set of previous predictions."""
    txt_mask = get_mask(sample_list["text_len"], sample_list["text"].size(1))  # :187-189
    txt_emb = text_bert(sd, cfg, sample_list["text"], txt_mask, train)  # :257-259
    if "text_bert_out_linear.weight" in sd:
        txt_emb = F.linear(txt_emb, sd["text_bert_out_linear.weight"], sd["text_bert_out_linear.bias"])  # :260
    (obj_in, obj_mask), (ocr_in, ocr_mask) = enc
    r = mmt(sd, cfg, txt_emb, txt_mask, obj_in, obj_mask, ocr_in, ocr_mask, sd["classifier.module.weight"], prev_inds, train)
    fixed = F.linear(r["mmt_dec_output"], sd["classifier.module.weight"], sd["classifier.module.bias"])  # :280
    dyn = ocr_ptr_net(sd, cfg, r["mmt_dec_output"], r["mmt_ocr_output"], ocr_mask)  # :281
    r["scores"] = torch.cat([fixed, dyn], dim=-1)  # :282
    return r


def m4c_forward(sd, cfg, sample_list, train=False, training_mode=True, return_all=False, relu_gates=None):
    """M4C.forward, m4c.py:171-181.  `training_mode` selects teacher forcing (`self.training`, :286-289) against the greedy
    decoding loop (:290-305); `train` only switches the dropouts on (never used for parity); `relu_gates` = {"obj": mask, "ocr": mask}, see `fc7`."""
    gates = relu_gates or {}
    enc = (obj_encoding(sd, cfg, sample_list, train, gates.get("obj")), ocr_encoding(sd, cfg, sample_list, train, gates.get("ocr")))
    if training_mode:
        r = forward_mmt_and_output(sd, cfg, sample_list, enc, sample_list["train_prev_inds"].clone(), train)
    else:
        prev = torch.zeros_like(sample_list["train_prev_inds"])
        prev[:, 0] = cfg.get("bos_idx", 1)  # :293-294
        for _ in range(prev.size(1)):  # :297-305
            r = forward_mmt_and_output(sd, cfg, sample_list, enc, prev, train)
            prev[:, 1:] = r["scores"].argmax(dim=-1)[:, :-1]
        r["prev_inds"] = prev
    if return_all:
        r["obj_mmt_in"], r["ocr_mmt_in"] = enc[0][0], enc[1][0]
        return r
    return {"scores": r["scores"]}


def decoding_bce_with_mask(scores, targets, loss_mask):
    """M4CDecodingBCEWithMaskLoss.forward, losses.py:581-592."""
    assert scores.dim() == 3 and loss_mask.dim() == 2
    losses = F.binary_cross_entropy_with_logits(scores, targets, reduction="none") * loss_mask.unsqueeze(-1)
    count = torch.max(torch.sum(loss_mask), torch.ones(1, device=scores.device))
    return torch.sum(losses) / count
