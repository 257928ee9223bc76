"""CPU oracle for MMF Transformer (`mmft`) — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

fp32 PyTorch restatement of mmf/models/mmf_transformer.py (`preprocess_sample` :180-401, `forward` :403-423),
mmf/models/transformers/base.py (`BaseTransformerBackend.forward` :327-342),
mmf/models/transformers/backends/huggingface.py (`HuggingfaceEmbeddings` :19-159, `generate_attention_mask` :204-210,
`generate_encoded_layers` :212-235) and mmf/models/transformers/heads/mlp.py (`MLP` :22-78), over the encoder
restated in oracle/visual_bert_oracle.py (`bert_layer`).

Parity status: PINNED against tests/golden/mmft_small64.npz, produced by running those reference classes
(tests/golden/make_golden.py::make_mmft).  Parameter names are the reference's (no `model.` prefix: MMFTransformer is
itself the BaseModel); the text modality ALIASES the transformer's word embedding table and LayerNorm
(huggingface.py:106-109).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from oracle.visual_bert_oracle import bert_layer, layer_norm

T_ = "backend.transformer."
E_ = "backend.embeddings."

DEFAULT_CONFIG = dict(
    vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
    max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
    attention_probs_dropout_prob=0.1, pad_token_id=0, num_labels=2, head_layer_norm_eps=1e-6, head_dropout_prob=0.1,
    modalities=[
        dict(type="text", key="text", segment_id=0, layer_norm_eps=1e-12, hidden_dropout_prob=0.1),
        dict(type="image", key="image", segment_id=1, embedding_dim=2048, layer_norm_eps=1e-12, hidden_dropout_prob=0.1),
    ],
)


def parameter_shapes(cfg):
    """Unique parameters (aliases listed once, under the transformer)."""
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    s = OrderedDict()
    e = T_ + "embeddings."
    s[e + "word_embeddings.weight"] = (cfg["vocab_size"], H)
    s[e + "position_embeddings.weight"] = (cfg["max_position_embeddings"], H)
    s[e + "token_type_embeddings.weight"] = (cfg["type_vocab_size"], H)
    s[e + "LayerNorm.weight"] = (H,)
    s[e + "LayerNorm.bias"] = (H,)
    for i in range(cfg["num_hidden_layers"]):
        p = T_ + "encoder.layer.%d." % i
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
    s[T_ + "pooler.dense.weight"] = (H, H)
    s[T_ + "pooler.dense.bias"] = (H,)
    for idx, m in enumerate(cfg["modalities"]):
        if m["type"] != "text":
            s[E_ + "token_embeddings.%d.0.weight" % idx] = (H, m["embedding_dim"])
            s[E_ + "token_embeddings.%d.0.bias" % idx] = (H,)
            s[E_ + "token_embeddings.%d.1.weight" % idx] = (H,)
            s[E_ + "token_embeddings.%d.1.bias" % idx] = (H,)
            s[E_ + "layer_norms.%d.weight" % idx] = (H,)
            s[E_ + "layer_norms.%d.bias" % idx] = (H,)
        s[E_ + "pos_embeddings.%d.weight" % idx] = (cfg["max_position_embeddings"], H)   # huggingface.py:111-114
    s[E_ + "token_type_embeddings.weight"] = (len(cfg["modalities"]), H)
    s["heads.0.pooler.dense.weight"] = (H, H)
    s["heads.0.pooler.dense.bias"] = (H,)
    s["heads.0.classifier.1.dense.weight"] = (H, H)
    s["heads.0.classifier.1.dense.bias"] = (H,)
    s["heads.0.classifier.1.LayerNorm.weight"] = (H,)
    s["heads.0.classifier.1.LayerNorm.bias"] = (H,)
    s["heads.0.classifier.2.weight"] = (cfg["num_labels"], H)
    s["heads.0.classifier.2.bias"] = (cfg["num_labels"],)
    return s


def shared(cfg):
    """Reference state-dict aliases created by huggingface.py:106-109."""
    out = {}
    for idx, m in enumerate(cfg["modalities"]):
        if m["type"] == "text":
            out[E_ + "token_embeddings.%d.weight" % idx] = T_ + "embeddings.word_embeddings.weight"
            out[E_ + "layer_norms.%d.weight" % idx] = T_ + "embeddings.LayerNorm.weight"
            out[E_ + "layer_norms.%d.bias" % idx] = T_ + "embeddings.LayerNorm.bias"
    return out


def preprocess_sample(cfg, sample_list):
    """MMFTransformer.preprocess_sample, mmf_transformer.py:180-360 (2-D text inputs, identity encoders)."""
    input_ids, position_ids, masks, segment_ids = {}, {}, {}, {}
    for m in cfg["modalities"]:
        key = m["key"]
        if m["type"] == "text":
            ids = sample_list["input_ids"] if "input_ids" in sample_list else sample_list[key]          # :232-241
        elif m["type"] == "image":
            ids = next(sample_list[k] for k in (key, "image", "input_modal", "image_feature_0") if k in sample_list)  # :242-245
        else:
            ids = sample_list[key]
        if m["type"] != "text" and ids.dim() == 2:
            ids = ids.unsqueeze(1)                                                                       # :253-254
        input_ids[key] = ids
        L = ids.size(1)
        position_ids[key] = torch.arange(0, L, dtype=torch.long, device=ids.device).unsqueeze(0).expand(ids.size(0), L)  # :277-289
        if m["type"] == "text" and "input_mask" in sample_list:
            masks[key] = sample_list["input_mask"]                                                       # :297-303
        elif key + "_mask" in sample_list:
            masks[key] = sample_list[key + "_mask"]                                                      # :305-307
        else:
            masks[key] = torch.ones(ids.shape[:2], dtype=torch.long, device=ids.device)                  # :309-313
        seg = m.get("segment_id", -1)
        if seg == -1:
            continue                                                                                     # :323-324
        if m["type"] == "text" and "segment_ids" in sample_list:
            segment_ids[key] = sample_list["segment_ids"]                                                # :325-333
        else:
            segment_ids[key] = torch.full(ids.shape[:2], fill_value=seg, dtype=torch.long, device=ids.device)  # :334-340
    return input_ids, position_ids, segment_ids, masks


def embeddings(sd, cfg, input_ids, position_ids, segment_ids, train=False):
    """HuggingfaceEmbeddings.forward, huggingface.py:131-159."""
    out = []
    for idx, m in enumerate(cfg["modalities"]):
        key = m["key"]
        eps = m.get("layer_norm_eps", cfg["layer_norm_eps"])
        if m["type"] == "text" and m.get("consume_raw", True):
            tok = F.embedding(input_ids[key], sd[T_ + "embeddings.word_embeddings.weight"], padding_idx=cfg.get("pad_token_id", 0))
            ln_w, ln_b = sd[T_ + "embeddings.LayerNorm.weight"], sd[T_ + "embeddings.LayerNorm.bias"]
        else:
            p = E_ + "token_embeddings.%d." % idx
            tok = layer_norm(F.linear(input_ids[key], sd[p + "0.weight"], sd[p + "0.bias"]), sd[p + "1.weight"], sd[p + "1.bias"], eps)  # :75-86
            ln_w, ln_b = sd[E_ + "layer_norms.%d.weight" % idx], sd[E_ + "layer_norms.%d.bias" % idx]
        total = tok
        if key in position_ids:
            total = total + F.embedding(position_ids[key], sd[E_ + "pos_embeddings.%d.weight" % idx])      # :149-150
        if key in segment_ids:
            total = total + F.embedding(segment_ids[key], sd[E_ + "token_type_embeddings.weight"])        # :152-155
        p_drop = m.get("hidden_dropout_prob", cfg["hidden_dropout_prob"]) if train else 0.0
        out.append(F.dropout(layer_norm(total, ln_w, ln_b, eps), p_drop, training=p_drop > 0))            # :157
    return torch.cat(out, dim=1)                                                                         # :159


def mlp_head(sd, cfg, sequence_output, train=False):
    """MLP.forward, heads/mlp.py:65-78 (bert_pooler, num_layers=1)."""
    pooled = torch.tanh(F.linear(sequence_output[:, 0], sd["heads.0.pooler.dense.weight"], sd["heads.0.pooler.dense.bias"]))
    p = cfg.get("head_dropout_prob", 0.1) if train else 0.0
    x = F.dropout(pooled, p, training=p > 0)
    x = F.gelu(F.linear(x, sd["heads.0.classifier.1.dense.weight"], sd["heads.0.classifier.1.dense.bias"]))
    x = layer_norm(x, sd["heads.0.classifier.1.LayerNorm.weight"], sd["heads.0.classifier.1.LayerNorm.bias"],
                   cfg.get("head_layer_norm_eps", 1e-6))
    logits = F.linear(x, sd["heads.0.classifier.2.weight"], sd["heads.0.classifier.2.bias"])
    return logits.view(-1, cfg["num_labels"])


def mmft_forward(sd, cfg, sample_list, train=False):
    """MMFTransformer.forward, mmf_transformer.py:403-423 + BaseTransformerBackend.forward, base.py:327-342."""
    input_ids, position_ids, segment_ids, masks = preprocess_sample(cfg, sample_list)
    attention_mask = torch.cat([masks[m["key"]] for m in cfg["modalities"]], dim=-1)                     # huggingface.py:206
    ext = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0                          # :207-208
    hidden = embeddings(sd, cfg, input_ids, position_ids, segment_ids, train)
    vb = {k.replace(T_ + "encoder.", "bert.encoder."): v for k, v in sd.items() if k.startswith(T_ + "encoder.")}
    hd = cfg["hidden_dropout_prob"] if train else 0.0
    ad = cfg["attention_probs_dropout_prob"] if train else 0.0
    for i in range(cfg["num_hidden_layers"]):
        hidden, _ = bert_layer(vb, cfg, i, hidden, ext, hd, ad)
    return {"scores": mlp_head(sd, cfg, hidden, train), "sequence_output": hidden}


def mlm_head(hsd, table, sequence_output, labels, eps=1e-5, ignore_index=-1):
    """MLM.forward, mmf/models/transformers/heads/mlm.py:49-97: the masked positions only (`sequence_output[masked_tokens, :]`, :80-83)
    through HF BertOnlyMLMHead (transform dense -> gelu -> LayerNorm(layer_norm_eps = 1e-5, :30), decoder tied to `table`, :45-46)
    and CrossEntropyLoss(ignore_index); NaN (nothing masked) becomes 0 (:89-94).  `hsd`: the head's parameters under `cls.*`."""
    masked = labels.ne(ignore_index)
    rows = sequence_output[masked, :]
    x = F.gelu(F.linear(rows, hsd["cls.predictions.transform.dense.weight"], hsd["cls.predictions.transform.dense.bias"]))
    x = F.layer_norm(x, (x.shape[-1],), hsd["cls.predictions.transform.LayerNorm.weight"], hsd["cls.predictions.transform.LayerNorm.bias"], eps)
    logits = F.linear(x, table, hsd["cls.predictions.bias"])
    loss = F.cross_entropy(logits.contiguous().view(-1, table.shape[0]), labels[masked].contiguous().view(-1), ignore_index=ignore_index)
    if torch.isnan(loss):
        loss = torch.nan_to_num(loss, nan=0.0)
    return {"logits": logits, "losses": {"masked_lm_loss": loss}}


def itm_head(hsd, sequence_output, is_correct, ignore_index=-1):
    """ITM.forward, mmf/models/transformers/heads/itm.py:40-74: HF BertPooler (tanh(dense(h[:, 0]))) -> BertOnlyNSPHead
    (`seq_relationship`: Linear(hidden, 2)) -> CrossEntropyLoss(ignore_index) against `itm_labels.is_correct`."""
    pooled = torch.tanh(F.linear(sequence_output[:, 0], hsd["pooler.dense.weight"], hsd["pooler.dense.bias"]))
    score = F.linear(pooled, hsd["cls.seq_relationship.weight"], hsd["cls.seq_relationship.bias"])
    loss = F.cross_entropy(score.contiguous().view(-1, 2), is_correct.contiguous().view(-1), ignore_index=ignore_index)
    return {"seq_relationship_score": score, "losses": {"itm_loss": loss}}


def mrc_head(hsd, sequence_output, region_class, region_mask, use_kl=True, eps=1e-12, ignore_index=-1):
    """MRC.forward, mmf/models/transformers/heads/mrc.py:47-90: the masked regions (`compute_masked_hidden`, heads/utils.py:169-179)
    through Linear -> GELU -> LayerNorm -> Linear(label_dim); KLDivLoss(batchmean) against the soft labels, or (use_kl = False)
    cross-entropy against their argmax over the non-background classes (:73-82)."""
    rows = sequence_output[region_mask.unsqueeze(-1).expand_as(sequence_output)].contiguous().view(-1, sequence_output.size(-1))
    x = F.gelu(F.linear(rows, hsd["region_classifier.0.weight"], hsd["region_classifier.0.bias"]))
    x = F.layer_norm(x, (x.shape[-1],), hsd["region_classifier.2.weight"], hsd["region_classifier.2.bias"], eps)
    pred = F.linear(x, hsd["region_classifier.3.weight"], hsd["region_classifier.3.bias"])
    if use_kl:
        loss = F.kl_div(F.log_softmax(pred, dim=-1), region_class, reduction="batchmean")
    else:
        loss = F.cross_entropy(pred, torch.max(region_class[:, 1:], dim=-1)[1] + 1, ignore_index=ignore_index, reduction="mean")
    return {"losses": {"mrc_loss": loss}}


def mrfr_head(hsd, img_embedding_weight, sequence_output, feat_targets, region_mask, eps=1e-12):
    """MRFR.forward (masked region feature regression), mmf/models/transformers/heads/mrfr.py:58-93: the masked regions through
    Linear -> GELU -> LayerNorm, projected back to the feature space with the TRANSPOSED image-embedding weight (`F.linear(h,
    W.t(), b)`, W = `img_embeddings.img_linear.weight` [hidden, img_dim], :46,86-88) and regressed onto the original features of the
    masked regions with mean-squared error."""
    rows = sequence_output[region_mask.unsqueeze(-1).expand_as(sequence_output)].contiguous().view(-1, sequence_output.size(-1))
    x = F.gelu(F.linear(rows, hsd["feat_regress.0.weight"], hsd["feat_regress.0.bias"]))
    x = F.layer_norm(x, (x.shape[-1],), hsd["feat_regress.2.weight"], hsd["feat_regress.2.bias"], eps)
    pred = F.linear(x, img_embedding_weight.t(), hsd["linear_proj_bias"])
    return {"losses": {"mrfr_loss": F.mse_loss(pred, feat_targets, reduction="mean")}}


def optimal_transport_dist(txt_emb, img_emb, txt_pad, img_pad, beta=0.5, iteration=50, k=1, eps=1e-5):
    """mmf/modules/ot.py:87-110 with `cost_matrix_cosine` (:15-25), `ipot` (:38-84, under no_grad: the transport plan is a constant of
    the backward pass) and `trace` (:28-35): the IPOT approximation of the optimal-transport distance between the text and the
    region embeddings of each sample under the cosine cost."""
    x = F.normalize(txt_emb, p=2, dim=-1, eps=eps)
    y = F.normalize(img_emb, p=2, dim=-1, eps=eps)
    cost = 1 - x.matmul(y.transpose(1, 2))                                           # [B, M, N]
    joint_pad = txt_pad.unsqueeze(-1) | img_pad.unsqueeze(-2)
    cost = cost.masked_fill(joint_pad, 0)
    x_len = (txt_pad.size(1) - txt_pad.sum(dim=1)).to(cost.dtype)
    y_len = (img_pad.size(1) - img_pad.sum(dim=1)).to(cost.dtype)
    with torch.no_grad():
        C = cost.detach()
        b, m, n = C.size()
        sigma = torch.ones(b, m, dtype=C.dtype) / x_len.unsqueeze(1)
        T = torch.ones(b, n, m, dtype=C.dtype)
        A = torch.exp(-C.transpose(1, 2) / beta)
        sigma = sigma.masked_fill(txt_pad, 0)
        jp = joint_pad.transpose(1, 2)
        T = T.masked_fill(jp, 0)
        A = A.masked_fill(jp, 0)
        xl, yl = x_len.view(b, 1, 1), y_len.view(b, 1, 1)
        x_mask = (txt_pad.to(C.dtype) * 1e4).unsqueeze(1)
        y_mask = (img_pad.to(C.dtype) * 1e4).unsqueeze(1)
        for _ in range(iteration):
            Q = A * T
            sigma = sigma.view(b, m, 1)
            for _ in range(k):
                delta = 1 / (yl * Q.matmul(sigma).view(b, 1, n) + y_mask)
                sigma = 1 / (xl * delta.matmul(Q) + x_mask)
            T = delta.view(b, n, 1) * Q * sigma
        T = T.masked_fill(jp, 0)
    prod = cost.matmul(T)                                                            # [B, M, M]
    return torch.diagonal(prod, dim1=1, dim2=2).sum(-1)


def wra_head(sequence_output, txt_len, img_len, txt_pad, img_pad, is_correct):
    """WRA.forward (word-region alignment), mmf/models/transformers/heads/wra.py:36-83: OT distance between the text rows [:tl] and
    the region rows [tl : tl + il] of the joint sequence; loss = (sum over matched pairs - sum over mismatched pairs) / number of
    pairs."""
    txt_emb = sequence_output[:, :txt_len, :]
    img_emb = sequence_output[:, txt_len:txt_len + img_len, :]
    ot = optimal_transport_dist(txt_emb.float(), img_emb.float(), txt_pad.bool(), img_pad.bool()).to(txt_emb)
    pos = ot.masked_select(is_correct == 1)
    neg = ot.masked_select(is_correct == 0)
    return {"losses": {"wra_loss": (pos.sum() - neg.sum()) / (pos.size(0) + neg.size(0))}, "ot_dist": ot}
