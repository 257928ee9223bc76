"""CPU oracle for UNITER (classification heads; SURVEY.md §8 a18 / f2, BASELINE.json configs[3]) — TEST INFRASTRUCTURE, NOT
PRODUCT CODE.

fp32 PyTorch restatement of mmf/models/uniter.py: `UNITERImageEmbeddings.forward` :69-87, `UNITERModelBase`
`_compute_txt_embeddings` :151-163, `_compute_img_embeddings` :165-178, `_compute_img_txt_embeddings` :180-196, `forward`
:198-246, `_infer_with_heads` :249-275, `UNITER.add_pos_feat` :686-717, `UNITER.add_custom_params` :719-745, and the `mlp`
head (mmf/models/transformers/heads/mlp.py:22-78), over the encoder restated in oracle/visual_bert_oracle.py.

Parity status: PINNED against tests/golden/uniter_small64.npz, produced by running those reference classes
(tests/golden/make_golden.py::make_uniter).  Parameter names are the reference's (`uniter.uniter.*`, `uniter.heads.<task>.*`).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from oracle.visual_bert_oracle import bert_layer, layer_norm

B_ = "uniter.uniter."

DEFAULT_CONFIG = dict(
    vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
    max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
    attention_probs_dropout_prob=0.1, pad_token_id=0, img_dim=2048, pos_dim=7, img_hidden_dropout_prob=0.0,
    task="vqa2", head_hidden_size=1536, head_layer_norm_eps=1e-6, head_dropout_prob=0.1, num_labels=3129,
)


def parameter_shapes(cfg):
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    s = OrderedDict()
    e = B_ + "text_embeddings."
    s[e + "word_embeddings.weight"] = (cfg["vocab_size"], H)
    s[e + "position_embeddings.weight"] = (cfg["max_position_embeddings"], H)
    s[e + "token_type_embeddings.weight"] = (cfg["type_vocab_size"], H)
    s[e + "LayerNorm.weight"] = (H,)
    s[e + "LayerNorm.bias"] = (H,)
    v = B_ + "img_embeddings."
    s[v + "img_linear.weight"] = (H, cfg["img_dim"])
    s[v + "img_linear.bias"] = (H,)
    s[v + "img_layer_norm.weight"] = (H,)
    s[v + "img_layer_norm.bias"] = (H,)
    s[v + "pos_layer_norm.weight"] = (H,)
    s[v + "pos_layer_norm.bias"] = (H,)
    s[v + "pos_linear.weight"] = (H, cfg.get("pos_dim", 7))
    s[v + "pos_linear.bias"] = (H,)
    s[v + "mask_embedding.weight"] = (2, cfg["img_dim"])
    s[v + "final_layer_norm.weight"] = (H,)
    s[v + "final_layer_norm.bias"] = (H,)
    for i in range(cfg["num_hidden_layers"]):
        p = B_ + "encoder.layer.%d." % i
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
    s[B_ + "pooler.dense.weight"] = (H, H)      # BertModel's pooler: kept by UNITERModelBase (:150), never called
    s[B_ + "pooler.dense.bias"] = (H,)
    h = "uniter.heads.%s." % cfg.get("task", "vqa2")
    HH = cfg["head_hidden_size"]
    s[h + "pooler.dense.weight"] = (H, H)
    s[h + "pooler.dense.bias"] = (H,)
    s[h + "classifier.1.dense.weight"] = (HH, H)
    s[h + "classifier.1.dense.bias"] = (HH,)
    s[h + "classifier.1.LayerNorm.weight"] = (HH,)
    s[h + "classifier.1.LayerNorm.bias"] = (HH,)
    s[h + "classifier.2.weight"] = (cfg["num_labels"], HH)
    s[h + "classifier.2.bias"] = (cfg["num_labels"],)
    return s


def pos_feat(bbox, image_width, image_height, like):
    """UNITER.add_pos_feat, uniter.py:686-717: (x1, y1, x2, y2, w, h, area), normalised by the image size when the
    boxes are in pixels (the reference's test: first coordinate >= 1 means "not normalised")."""
    norm_xy = torch.as_tensor(bbox)[:, :, :4].clone().float()
    if norm_xy[0, 0, 0] < 1:      # :697 — the reference divides when the FIRST coordinate is below 1 (sic)
        img_h = torch.as_tensor(image_height).unsqueeze(1).unsqueeze(1)
        img_w = torch.as_tensor(image_width).unsqueeze(1).unsqueeze(1)
        norm_xy = norm_xy / torch.cat([img_w, img_h, img_w, img_h], dim=-1).to(norm_xy.device)
    bbox_w = (norm_xy[:, :, 2] - norm_xy[:, :, 0]).unsqueeze(-1)
    bbox_h = (norm_xy[:, :, 3] - norm_xy[:, :, 1]).unsqueeze(-1)
    return torch.cat([norm_xy, bbox_w, bbox_h, bbox_w * bbox_h], dim=-1).to(like)


def prepare_inputs(sample_list):
    """UNITER.add_custom_params, uniter.py:719-745."""
    feats = sample_list["image_feature_0"]
    info = sample_list["image_info_0"]
    image_dim = info["max_features"]
    image_mask = torch.arange(feats.size(-2), device=feats.device).expand(feats.size()[:-1])
    if image_dim.dim() < image_mask.dim():
        image_dim = image_dim.unsqueeze(-1)
    image_mask = (image_mask < image_dim).long()
    return dict(input_ids=sample_list["input_ids"], image_feat=feats, image_mask=image_mask,
                attention_mask=torch.cat((sample_list["input_mask"], image_mask), dim=-1),
                position_ids=torch.arange(0, sample_list["input_ids"].size(1), dtype=torch.long, device=feats.device).unsqueeze(0),
                img_pos_feat=pos_feat(info["bbox"], info.get("image_width"), info.get("image_height"), feats))


def image_embeddings(sd, cfg, img_feat, img_pos_feat, type_embeddings, img_masks, train=False):
    """UNITERImageEmbeddings.forward, uniter.py:69-87."""
    v = B_ + "img_embeddings."
    eps = 1e-12
    if img_masks is not None:
        table = sd[v + "mask_embedding.weight"]
        table = torch.cat([torch.zeros_like(table[:1]), table[1:]], dim=0)     # :76 row 0 is reset to zero every call
        img_feat = img_feat + F.embedding(img_masks.long(), table, padding_idx=0)   # :77-78
    t_im = layer_norm(F.linear(img_feat, sd[v + "img_linear.weight"], sd[v + "img_linear.bias"]),
                      sd[v + "img_layer_norm.weight"], sd[v + "img_layer_norm.bias"], eps)            # :80
    t_pos = layer_norm(F.linear(img_pos_feat, sd[v + "pos_linear.weight"], sd[v + "pos_linear.bias"]),
                       sd[v + "pos_layer_norm.weight"], sd[v + "pos_layer_norm.bias"], eps)           # :81
    emb = layer_norm(t_im + t_pos + type_embeddings, sd[v + "final_layer_norm.weight"], sd[v + "final_layer_norm.bias"], eps)  # :82-83
    p = cfg.get("img_hidden_dropout_prob", 0.0) if train else 0.0
    return F.dropout(emb, p, training=p > 0)


def model_base(sd, cfg, inp, train=False):
    """UNITERModelBase.forward, uniter.py:198-246 (input_modality "image-text")."""
    e = B_ + "text_embeddings."
    ids = inp["input_ids"]
    hd = cfg["hidden_dropout_prob"] if train else 0.0
    ad = cfg["attention_probs_dropout_prob"] if train else 0.0
    ext = (1.0 - inp["attention_mask"][:, None, None, :].to(torch.float32)) * -10000.0                 # :210-217
    txt = (F.embedding(ids, sd[e + "word_embeddings.weight"], padding_idx=cfg.get("pad_token_id", 0))
           + F.embedding(torch.zeros_like(ids), sd[e + "token_type_embeddings.weight"])
           + F.embedding(inp["position_ids"].expand(ids.shape), sd[e + "position_embeddings.weight"]))  # HF BertEmbeddings
    txt = F.dropout(layer_norm(txt, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], cfg["layer_norm_eps"]), hd, training=hd > 0)
    img_type_ids = torch.ones_like(inp["image_feat"][:, :, 0].long())                                  # :172-173
    type_emb = F.embedding(img_type_ids, sd[e + "token_type_embeddings.weight"])                       # :174
    img = image_embeddings(sd, cfg, inp["image_feat"], inp["img_pos_feat"], type_emb, inp["image_mask"], train)
    hidden = torch.cat([txt, img], dim=1)                                                              # :195
    vb = {k.replace(B_ + "encoder.", "bert.encoder."): v for k, v in sd.items() if k.startswith(B_ + "encoder.")}
    for i in range(cfg["num_hidden_layers"]):
        hidden, _ = bert_layer(vb, cfg, i, hidden, ext, hd, ad)
    return hidden


def uniter_forward(sd, cfg, sample_list, train=False):
    """UNITER.forward :747-749 -> UNITERForClassification.forward -> _infer_with_heads :249-275 with the `mlp` head."""
    inp = prepare_inputs(sample_list)
    seq = model_base(sd, cfg, inp, train)
    h = "uniter.heads.%s." % cfg.get("task", "vqa2")
    pooled = torch.tanh(F.linear(seq[:, 0], sd[h + "pooler.dense.weight"], sd[h + "pooler.dense.bias"]))
    p = cfg.get("head_dropout_prob", 0.1) if train else 0.0
    x = F.dropout(pooled, p, training=p > 0)
    x = F.gelu(F.linear(x, sd[h + "classifier.1.dense.weight"], sd[h + "classifier.1.dense.bias"]))
    x = layer_norm(x, sd[h + "classifier.1.LayerNorm.weight"], sd[h + "classifier.1.LayerNorm.bias"], cfg.get("head_layer_norm_eps", 1e-6))
    logits = F.linear(x, sd[h + "classifier.2.weight"], sd[h + "classifier.2.bias"])
    return {"scores": logits.contiguous().view(-1, logits.size(-1)), "sequence_output": seq, "img_pos_feat": inp["img_pos_feat"]}


def pretraining_preprocess(sample_list, task, region_masks=None, ignore_index=-1):
    """UNITERForPretraining's host-side preparation for one task, uniter.py:411-617: `_process_sample_list_for_pretraining`
    (:449-467; `_remove_mismatched_captions`, :583-617, never writes its selection back — a no-op, kept as one),
    `_add_image_feat_masked` (:469-478: the regions of the random mask `region_masks` — drawn by `_get_img_mask`, :480-485, here an
    INPUT — are zeroed, and that mask replaces `image_mask`), `_preprocess_mlm` (:487-504), `_preprocess_itm` (:506-511),
    `_preprocess_mrc` (:528-544) + `_mask_inputs_in_sample_list` (:519-526).  Returns a new dict."""
    s = dict(sample_list)
    if task in ("mrfr", "mrc"):
        m = region_masks.bool()
        s["image_feat_masked"] = s["image_feat"].masked_fill(m.unsqueeze(-1).expand_as(s["image_feat"]), 0)      # :473-476
        s["image_mask"] = m                                                                                        # :477
        s["cls_prob"] = torch.as_tensor(s["image_info_0"]["cls_prob"])                                             # :458-460
    if task == "mlm":
        text = s["lm_label_ids"]
        image = torch.full(s["image_feat"].shape[:2], fill_value=ignore_index, dtype=torch.long)
        s["mlm_labels"] = {"text": text, "image": image, "combined_labels": torch.cat([text, image], dim=-1)}      # :492-503
        s["input_ids"] = s["input_ids_masked"]                                                                     # :504
    elif task == "itm":
        s["itm_labels"] = {"is_correct": s["is_correct"]}
    elif task == "mrc":
        m = s["image_mask"]
        cls_prob = s["cls_prob"]
        s["region_class"] = cls_prob[m.unsqueeze(-1).expand_as(cls_prob)].contiguous().view(-1, cls_prob.size(2))  # :536-541
        pad = torch.zeros((m.size(0), s["input_ids"].size(1))).to(m)
        s["image_region_mask"] = torch.cat([pad, m], dim=-1)                                                       # :513-517, 523-525
        s["image_feat"] = s["image_feat_masked"]                                                                   # :526
    elif task == "mrfr":                                                                                           # :542-556
        m = s["image_mask"]
        feat = s["image_feat"]
        s["mrfr_region_target"] = feat[m.unsqueeze(-1).expand_as(feat)].contiguous().view(-1, feat.size(2))      # the ORIGINAL features
        pad = torch.zeros((m.size(0), s["input_ids"].size(1))).to(m)
        s["mrfr_region_mask"] = torch.cat([pad, m], dim=-1)
        s["image_feat"] = s["image_feat_masked"]
    elif task == "wra":                                                                                            # :558-581 (dense batch: no padding)
        B = s["input_ids"].size(0)
        s["wra_info"] = {"txt_pad": torch.zeros(B, s["input_ids"].size(1), dtype=torch.bool),
                         "img_pad": torch.zeros(B, s["image_feat"].size(1), dtype=torch.bool)}
    else:
        raise ValueError("Task %s is not restated (mlm, itm, mrc, mrfr, wra)" % task)
    return s


def uniter_pretraining_forward(sd, cfg, sample_list, task, region_masks=None):
    """UNITERForPretraining.forward (uniter.py:411-440) -> `_infer_with_heads` (:249-275) for task in {mlm, itm, mrc, mrfr, wra}: the heads'
    own restatements are oracle.mmft_oracle.{mlm_head, itm_head, mrc_head, mrfr_head, wra_head}.  The MLM decoder is NOT tied here (the reference's
    pretraining wrapper builds the head without calling `tie_weights`): `heads.mlm.cls.predictions.decoder.weight` is its own tensor."""
    from oracle import mmft_oracle as H
    s = pretraining_preprocess(sample_list, task, region_masks)
    seq = model_base(sd, cfg, s)                                             # (img_masks = s["image_mask"], uniter.py:262)
    pre = "uniter.heads.%s." % task
    hsd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    if task == "mlm":
        out = H.mlm_head(hsd, hsd["cls.predictions.decoder.weight"], seq, s["mlm_labels"]["combined_labels"])
    elif task == "itm":
        out = H.itm_head(hsd, seq, s["itm_labels"]["is_correct"])
    elif task == "mrfr":     # the head's projection weight is the image embedding's (uniter.py:397-400), applied transposed
        out = H.mrfr_head(hsd, sd["uniter.uniter.img_embeddings.img_linear.weight"], seq, s["mrfr_region_target"], s["mrfr_region_mask"].bool())
    elif task == "wra":
        out = H.wra_head(seq, s["input_ids"].size(1), s["image_feat"].size(1), s["wra_info"]["txt_pad"], s["wra_info"]["img_pad"], s["is_correct"])
    else:
        out = H.mrc_head(hsd, seq, s["region_class"], s["image_region_mask"].bool())
    out["preprocessed"] = s
    return out
