"""CPU oracle for ViLBERT (BASELINE.json configs[2]) — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

fp32 PyTorch restatement of mmf/models/vilbert.py: `BertSelfAttention` :46-114 / `BertImageSelfAttention` :153-247
(dynamic_attention=False), `BertLayer` :131-146 / `BertImageLayer` :313-332, `BertBiAttention` :347-475, `BertBiOutput`
:478-512, `BertConnectionLayer` :515-556, `BertEncoder.forward` :590-796 (fixed_{t,v}_layer :625-666, in_batch_pairs / fast mode :678-725),
`BertTextPooler` / `BertImagePooler` :799-826, `BertImageFeatureEmbeddings` :891-913, `ViLBERTBase.forward` :936-1051,
`ViLBERTForClassification.forward` :1281-1333 and `ViLBERT.forward` :1364-1446 (input massaging).

Parity status: PINNED against tests/golden/vilbert_small.npz, produced by running those reference classes
(tests/golden/make_golden.py::make_vilbert; text heads d=64, visual and co-attention heads d=128 as in the real config), and the variants
against vilbert_dyn / vilbert_fixed / vilbert_pairs (in_batch_pairs) / vilbert_fast (fast_mode) / vilbert_nlvr2 / vilbert_pretraining*.npz.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from oracle.visual_bert_oracle import layer_norm

DEFAULT_CONFIG = dict(
    vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
    max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
    attention_probs_dropout_prob=0.1, pad_token_id=0, v_feature_size=2048, v_hidden_size=1024, v_num_hidden_layers=6,
    v_num_attention_heads=8, v_intermediate_size=1024, bi_hidden_size=1024, bi_num_attention_heads=8,
    v_attention_probs_dropout_prob=0.1, v_hidden_dropout_prob=0.1, v_biattention_id=[0, 1, 2, 3, 4, 5],
    t_biattention_id=[6, 7, 8, 9, 10, 11], fusion_method="mul", num_labels=3129,
)


def _layer_shapes(s, p, H, I):
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
    H, I, VH, VI, BH = (cfg["hidden_size"], cfg["intermediate_size"], cfg["v_hidden_size"], cfg["v_intermediate_size"],
                        cfg["bi_hidden_size"])
    s = OrderedDict()
    e = "bert.embeddings."
    s[e + "word_embeddings.weight"] = (cfg["vocab_size"], H)
    s[e + "position_embeddings.weight"] = (cfg["max_position_embeddings"], H)
    s[e + "token_type_embeddings.weight"] = (cfg["type_vocab_size"], H)
    s[e + "LayerNorm.weight"] = (H,)
    s[e + "LayerNorm.bias"] = (H,)
    v = "bert.v_embeddings."
    s[v + "image_embeddings.weight"] = (VH, cfg["v_feature_size"])
    s[v + "image_embeddings.bias"] = (VH,)
    s[v + "image_location_embeddings.weight"] = (VH, 5)
    s[v + "image_location_embeddings.bias"] = (VH,)
    s[v + "LayerNorm.weight"] = (VH,)
    s[v + "LayerNorm.bias"] = (VH,)
    for i in range(cfg["num_hidden_layers"]):
        _layer_shapes(s, "bert.encoder.layer.%d." % i, H, I)
    for i in range(cfg["v_num_hidden_layers"]):
        _layer_shapes(s, "bert.encoder.v_layer.%d." % i, VH, VI)
        if cfg.get("dynamic_attention", False):      # vilbert.py:174-176: gates fed from the text stream's width
            for n in ("dyLinear_q", "dyLinear_k"):
                s["bert.encoder.v_layer.%d.attention.self.%s.weight" % (i, n)] = (VH, H)
                s["bert.encoder.v_layer.%d.attention.self.%s.bias" % (i, n)] = (VH,)
    for i in range(len(cfg["v_biattention_id"])):
        p = "bert.encoder.c_layer.%d." % i
        for n, din in (("query1", VH), ("key1", VH), ("value1", VH), ("query2", H), ("key2", H), ("value2", H)):
            s[p + "biattention.%s.weight" % n] = (BH, din)
            s[p + "biattention.%s.bias" % n] = (BH,)
        for n, dout in (("1", VH), ("2", H)):
            if n == "2":
                pass
            s[p + "biOutput.dense%s.weight" % n] = (dout, BH)
            s[p + "biOutput.dense%s.bias" % n] = (dout,)
            s[p + "biOutput.LayerNorm%s.weight" % n] = (dout,)
            s[p + "biOutput.LayerNorm%s.bias" % n] = (dout,)
            s[p + "biOutput.q_dense%s.weight" % n] = (dout, BH)      # declared, never used (vilbert.py:486,493)
            s[p + "biOutput.q_dense%s.bias" % n] = (dout,)
        s[p + "v_intermediate.dense.weight"] = (VI, VH)
        s[p + "v_intermediate.dense.bias"] = (VI,)
        s[p + "v_output.dense.weight"] = (VH, VI)
        s[p + "v_output.dense.bias"] = (VH,)
        s[p + "v_output.LayerNorm.weight"] = (VH,)
        s[p + "v_output.LayerNorm.bias"] = (VH,)
        s[p + "t_intermediate.dense.weight"] = (I, H)
        s[p + "t_intermediate.dense.bias"] = (I,)
        s[p + "t_output.dense.weight"] = (H, I)
        s[p + "t_output.dense.bias"] = (H,)
        s[p + "t_output.LayerNorm.weight"] = (H,)
        s[p + "t_output.LayerNorm.bias"] = (H,)
    s["bert.t_pooler.dense.weight"] = (BH, H)
    s["bert.t_pooler.dense.bias"] = (BH,)
    s["bert.v_pooler.dense.weight"] = (BH, VH)
    s["bert.v_pooler.dense.bias"] = (BH,)
    if cfg.get("training_head_type", "classification") == "pretraining":
        # ViLBERTForPretraining (vilbert.py:1054-1077): `cls` = vilbert.BertPreTrainingHeads (:861-868).  `cls.predictions.decoder.weight`
        # is the word-embedding table (:1088-1095) and `.decoder.bias` is `cls.predictions.bias` (HF <= 4.10 BertLMPredictionHead)
        s["cls.predictions.bias"] = (cfg["vocab_size"],)
        s["cls.predictions.transform.dense.weight"] = (H, H)
        s["cls.predictions.transform.dense.bias"] = (H,)
        s["cls.predictions.transform.LayerNorm.weight"] = (H,)
        s["cls.predictions.transform.LayerNorm.bias"] = (H,)
        s["cls.bi_seq_relationship.weight"] = (2, BH)
        s["cls.bi_seq_relationship.bias"] = (2,)
        s["cls.imagePredictions.transform.dense.weight"] = (VH, VH)
        s["cls.imagePredictions.transform.dense.bias"] = (VH,)
        s["cls.imagePredictions.transform.LayerNorm.weight"] = (VH,)
        s["cls.imagePredictions.transform.LayerNorm.bias"] = (VH,)
        s["cls.imagePredictions.decoder.weight"] = (cfg["v_target_size"], VH)
        s["cls.imagePredictions.decoder.bias"] = (cfg["v_target_size"],)
        return s
    if cfg.get("training_head_type", "classification") == "nlvr2":
        BH = 2 * BH        # vilbert.py:1262-1265: the head runs on pairs of pooled vectors
    s["classifier.0.dense.weight"] = (BH, BH)
    s["classifier.0.dense.bias"] = (BH,)
    s["classifier.0.LayerNorm.weight"] = (BH,)
    s["classifier.0.LayerNorm.bias"] = (BH,)
    s["classifier.1.weight"] = (cfg["num_labels"], BH)
    s["classifier.1.bias"] = (cfg["num_labels"],)
    return s


def _heads(x, n):
    B, S, Hd = x.shape
    return x.view(B, S, n, Hd // n).permute(0, 2, 1, 3)   # transpose_for_scores, :66-72


def attention(q, k, v, n_heads, ext_mask, dropout_p):
    """scores = Q K^T / sqrt(d) + mask ; softmax ; dropout ; P V ; merge heads   (:89-104, :217-236, :418-452)."""
    ql, kl, vl = _heads(q, n_heads), _heads(k, n_heads), _heads(v, n_heads)
    scores = torch.matmul(ql, kl.transpose(-1, -2)) / math.sqrt(ql.shape[-1])
    scores = scores + ext_mask
    probs = F.dropout(F.softmax(scores, dim=-1), dropout_p, training=dropout_p > 0)
    ctx = torch.matmul(probs, vl).permute(0, 2, 1, 3).contiguous()
    return ctx.view(ctx.shape[0], ctx.shape[1], -1)


def stream_layer(sd, p, n_heads, hidden, ext_mask, hidden_dropout, attn_dropout, eps=1e-12, txt=None, txt_mask2=None):
    """BertLayer.forward :138-146 (text) and BertImageLayer.forward :320-332 (visual): the HF BertSelfOutput / BertIntermediate /
    BertOutput blocks and their Image twins (:250-310) are the same arithmetic.  `txt` / `txt_mask2` ([B, T, H_t], [B, T, 1]):
    BertImageSelfAttention with dynamic_attention, :199-212 — queries and keys gated by 1 + sigmoid(Linear(masked text mean))."""
    lin = lambda x, n: F.linear(x, sd[p + n + ".weight"], sd[p + n + ".bias"])
    q, k = lin(hidden, "attention.self.query"), lin(hidden, "attention.self.key")
    if txt is not None:
        pool = (txt * txt_mask2).sum(1) / txt_mask2.sum(1)                               # :204-205
        q = q * (1 + torch.sigmoid(lin(pool, "attention.self.dyLinear_q"))).unsqueeze(1)      # :208, :211
        k = k * (1 + torch.sigmoid(lin(pool, "attention.self.dyLinear_k"))).unsqueeze(1)      # :209, :212
    ctx = attention(q, k, lin(hidden, "attention.self.value"),
                    n_heads, ext_mask, attn_dropout)
    a = F.dropout(lin(ctx, "attention.output.dense"), hidden_dropout, training=hidden_dropout > 0)
    a = layer_norm(a + hidden, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], eps)
    h = F.gelu(lin(a, "intermediate.dense"))
    o = F.dropout(lin(h, "output.dense"), hidden_dropout, training=hidden_dropout > 0)
    return layer_norm(o + a, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)


def connection_layer(sd, cfg, i, img, img_mask, txt, txt_mask, train):
    """BertConnectionLayer.forward :528-556 = BertBiAttention :388-475 + BertBiOutput :496-512 + the two FFNs."""
    p = "bert.encoder.c_layer.%d." % i
    lin = lambda x, n: F.linear(x, sd[p + n + ".weight"], sd[p + n + ".bias"])
    nh = cfg["bi_num_attention_heads"]
    vd = cfg["v_hidden_dropout_prob"] if train else 0.0
    td = cfg["hidden_dropout_prob"] if train else 0.0
    vad = cfg["v_attention_probs_dropout_prob"] if train else 0.0
    tad = cfg["attention_probs_dropout_prob"] if train else 0.0
    # text queries over image keys/values -> context_layer1 (goes to the TEXT stream); mask = image mask (:418-436)
    ctx1 = attention(lin(txt, "biattention.query2"), lin(img, "biattention.key1"), lin(img, "biattention.value1"), nh, img_mask, vad)
    # image queries over text keys/values -> context_layer2 (goes to the IMAGE stream); mask = text mask (:438-457)
    ctx2 = attention(lin(img, "biattention.query1"), lin(txt, "biattention.key2"), lin(txt, "biattention.value2"), nh, txt_mask, tad)
    # biOutput(bi_output2, input_tensor1, bi_output1, input_tensor2), :538-540
    a1 = F.dropout(lin(ctx2, "biOutput.dense1"), vd, training=vd > 0)
    a2 = F.dropout(lin(ctx1, "biOutput.dense2"), td, training=td > 0)
    a1 = layer_norm(a1 + img, sd[p + "biOutput.LayerNorm1.weight"], sd[p + "biOutput.LayerNorm1.bias"], 1e-12)
    a2 = layer_norm(a2 + txt, sd[p + "biOutput.LayerNorm2.weight"], sd[p + "biOutput.LayerNorm2.bias"], 1e-12)
    h1 = F.gelu(lin(a1, "v_intermediate.dense"))
    o1 = F.dropout(lin(h1, "v_output.dense"), vd, training=vd > 0)
    o1 = layer_norm(o1 + a1, sd[p + "v_output.LayerNorm.weight"], sd[p + "v_output.LayerNorm.bias"], 1e-12)
    h2 = F.gelu(lin(a2, "t_intermediate.dense"))
    o2 = F.dropout(lin(h2, "t_output.dense"), td, training=td > 0)
    o2 = layer_norm(o2 + a2, sd[p + "t_output.LayerNorm.weight"], sd[p + "t_output.LayerNorm.bias"], cfg["layer_norm_eps"])
    return o1, o2


def encoder(sd, cfg, txt, img, txt_mask, img_mask, train=False, txt_mask2=None):
    """BertEncoder.forward :590-796, with_coattention, no batch expansion.  `fixed_t_layer` / `fixed_v_layer` (:625-666): the reference runs
    `forward_no_grad` on the layer at t_start and sets t_start = fixed_t_layer inside that very loop iteration, so only the FIRST such layer
    executes (detached) and layers t_start + 1 .. fixed_t_layer - 1 are never run; restated as it behaves."""
    td = cfg["hidden_dropout_prob"] if train else 0.0
    tad = cfg["attention_probs_dropout_prob"] if train else 0.0
    vd = cfg["v_hidden_dropout_prob"] if train else 0.0
    vad = cfg["v_attention_probs_dropout_prob"] if train else 0.0
    t_layer = lambda i, x: stream_layer(sd, "bert.encoder.layer.%d." % i, cfg["num_attention_heads"], x, txt_mask, td, tad,
                                        cfg["layer_norm_eps"])
    dyn = bool(cfg.get("dynamic_attention", False))
    # (the visual layers see the text stream as it is when they run, :667-673)
    v_layer = lambda i, x, t: stream_layer(sd, "bert.encoder.v_layer.%d." % i, cfg["v_num_attention_heads"], x, img_mask, vd, vad,
                                           txt=t if dyn else None, txt_mask2=txt_mask2)
    v_start = t_start = 0
    fixed_t, fixed_v = int(cfg.get("fixed_t_layer", 0)), int(cfg.get("fixed_v_layer", 0))
    for count, (v_end, t_end) in enumerate(zip(cfg["v_biattention_id"], cfg["t_biattention_id"])):
        assert fixed_t <= t_end and fixed_v <= v_end                                  # :622-623
        if t_start < fixed_t:                                                         # :626-634
            with torch.no_grad():
                txt = t_layer(t_start, txt)
            t_start = fixed_t
        for i in range(t_start, t_end):
            txt = t_layer(i, txt)
        if v_start < fixed_v:                                                         # :646-661
            with torch.no_grad():
                img = v_layer(v_start, img, txt)
            v_start = fixed_v
        for i in range(v_start, v_end):
            img = v_layer(i, img, txt)
        if count == 0 and cfg.get("in_batch_pairs", False):                           # :678-710, batch_size ^ 2 pairs (text i, image j)
            B = txt.shape[0]
            img = img.unsqueeze(0).expand(B, *img.shape).reshape(B * B, *img.shape[1:])
            img_mask = img_mask.unsqueeze(0).expand(B, *img_mask.shape).reshape(B * B, *img_mask.shape[1:])
            txt = txt.unsqueeze(1).expand(B, B, *txt.shape[1:]).reshape(B * B, *txt.shape[1:])
            txt_mask = txt_mask.unsqueeze(1).expand(B, B, *txt_mask.shape[1:]).reshape(B * B, *txt_mask.shape[1:])
            t_layer = lambda i, x, m=txt_mask: stream_layer(sd, "bert.encoder.layer.%d." % i, cfg["num_attention_heads"], x, m, td, tad, cfg["layer_norm_eps"])
            v_layer = lambda i, x, t, m=img_mask: stream_layer(sd, "bert.encoder.v_layer.%d." % i, cfg["v_num_attention_heads"], x, m, vd, vad)
        if count == 0 and cfg.get("fast_mode", False) and txt.shape[0] != img.shape[0]:    # :712-723
            txt = txt.expand(img.shape[0], *txt.shape[1:])
            txt_mask = txt_mask.expand(img.shape[0], *txt_mask.shape[1:])
            t_layer = lambda i, x, m=txt_mask: stream_layer(sd, "bert.encoder.layer.%d." % i, cfg["num_attention_heads"], x, m, td, tad, cfg["layer_norm_eps"])
        img, txt = connection_layer(sd, cfg, count, img, img_mask, txt, txt_mask, train)
        v_start, t_start = v_end, t_end
    for i in range(v_start, cfg["v_num_hidden_layers"]):
        img = v_layer(i, img, txt)
    for i in range(t_start, cfg["num_hidden_layers"]):
        txt = t_layer(i, txt)
    return txt, img


def vilbert_base(sd, cfg, input_txt, image_feature, image_location, token_type_ids, attention_mask, image_attention_mask,
                 train=False, pooler_masks=None):
    """ViLBERTBase.forward :936-1051.

    `pooler_masks=(mask_t, mask_v)` (test aid) evaluates the two ReLU poolers on a GIVEN branch (`pre * mask` instead of
    `relu(pre)`): a pre-activation within bf16 noise of zero can land on either side of the kink in a reduced-precision
    run, and the gradient of a piecewise-linear map is only comparable on the same piece."""
    if attention_mask is None:
        attention_mask = torch.ones_like(input_txt)
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_txt)
    if image_attention_mask is None:
        image_attention_mask = torch.ones(image_feature.size(0), image_feature.size(1)).type_as(input_txt)
    ext_t = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0            # :988-1000
    ext_v = (1.0 - image_attention_mask[:, None, None, :].to(torch.float32)) * -10000.0      # :1009
    hd = cfg["hidden_dropout_prob"] if train else 0.0
    e = "bert.embeddings."
    T = input_txt.size(1)
    pos = torch.arange(T, device=input_txt.device).unsqueeze(0).expand(input_txt.shape)
    txt = (F.embedding(input_txt, sd[e + "word_embeddings.weight"], padding_idx=cfg.get("pad_token_id", 0))
           + F.embedding(token_type_ids, sd[e + "token_type_embeddings.weight"])
           + F.embedding(pos, sd[e + "position_embeddings.weight"]))                          # HF BertEmbeddings
    txt = F.dropout(layer_norm(txt, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], cfg["layer_norm_eps"]), hd, training=hd > 0)
    v = "bert.v_embeddings."
    img = (F.linear(image_feature, sd[v + "image_embeddings.weight"], sd[v + "image_embeddings.bias"])
           + F.linear(image_location, sd[v + "image_location_embeddings.weight"], sd[v + "image_location_embeddings.bias"]))  # :905-910
    img = F.dropout(layer_norm(img, sd[v + "LayerNorm.weight"], sd[v + "LayerNorm.bias"], 1e-12), hd, training=hd > 0)     # :911
    seq_t, seq_v = encoder(sd, cfg, txt, img, ext_t, ext_v, train,
                           txt_mask2=attention_mask.unsqueeze(2).to(torch.float32))          # extended_attention_mask2, :985
    pre_t = F.linear(seq_t[:, 0], sd["bert.t_pooler.dense.weight"], sd["bert.t_pooler.dense.bias"])   # :805-811
    pre_v = F.linear(seq_v[:, 0], sd["bert.v_pooler.dense.weight"], sd["bert.v_pooler.dense.bias"])   # :820-826
    pooled_t = F.relu(pre_t) if pooler_masks is None else pre_t * pooler_masks[0]
    pooled_v = F.relu(pre_v) if pooler_masks is None else pre_v * pooler_masks[1]
    return seq_t, seq_v, pooled_t, pooled_v


def prepare_inputs(sample_list):
    """ViLBERT.get_image_and_text_features :1364-1418 (non-nlvr2) + the mask of :1431-1443."""
    ids, amask, tt = sample_list["input_ids"], sample_list["input_mask"], sample_list["segment_ids"]
    if sample_list.get("dataset_name", None) == "nlvr2":      # :1369-1394: text repeated, the two images stacked
        ids, amask, tt = torch.cat([ids, ids]), torch.cat([amask, amask]), torch.cat([tt, tt])
        i0, i1 = sample_list["img0"], sample_list["img1"]
        feats = torch.cat([i0["image_feature_0"], i1["image_feature_0"]])
        info = {"max_features": torch.cat([i0["image_info_0"]["max_features"], i1["image_info_0"]["max_features"]]),
                "bbox": torch.cat([i0["image_info_0"]["bbox"], i1["image_info_0"]["bbox"]])}
    else:
        info = sample_list.get("image_info_0", None) or {}
        feats = sample_list["image_feature_0"]
    image_dim = info.get("max_features", None)
    image_mask = None
    if feats is not None and image_dim is not None:
        image_mask = torch.arange(feats.size(-2), device=feats.device).expand(*feats.size()[:-1])
        if image_dim.dim() < image_mask.dim():
            image_dim = image_dim.unsqueeze(-1)
        image_mask = (image_mask < image_dim).long()
    return dict(input_ids=ids, attention_mask=amask, token_type_ids=tt, image_feature=feats, image_location=info.get("bbox", None),
                image_attention_mask=image_mask)


def vilbert_forward(sd, cfg, sample_list, train=False, pooler_masks=None):
    """ViLBERT.forward :1423-1446 -> ViLBERTForClassification.forward :1281-1333."""
    p = prepare_inputs(sample_list)
    seq_t, seq_v, pooled_t, pooled_v = vilbert_base(sd, cfg, p["input_ids"], p["image_feature"], p["image_location"],
                                                    p["token_type_ids"], p["attention_mask"], p["image_attention_mask"], train,
                                                    pooler_masks)
    fused = pooled_t * pooled_v if cfg.get("fusion_method", "mul") == "mul" else pooled_t + pooled_v   # :1315-1320
    hd = cfg["hidden_dropout_prob"] if train else 0.0
    x = F.dropout(fused, hd, training=hd > 0)
    if cfg.get("training_head_type", "classification") == "nlvr2":
        x = x.view(-1, x.size(1) * 2)        # :1322-1323 (pairs CONSECUTIVE rows of the stacked batch, as the reference does)
    x = F.gelu(F.linear(x, sd["classifier.0.dense.weight"], sd["classifier.0.dense.bias"]))
    x = layer_norm(x, sd["classifier.0.LayerNorm.weight"], sd["classifier.0.LayerNorm.bias"], cfg["layer_norm_eps"])
    logits = F.linear(x, sd["classifier.1.weight"], sd["classifier.1.bias"])
    return {"scores": logits.contiguous().view(-1, cfg["num_labels"]), "sequence_output_t": seq_t, "sequence_output_v": seq_v,
            "pooled_output_t": pooled_t, "pooled_output_v": pooled_v}


def vilbert_pretraining_forward(sd, cfg, sample_list, train=False):
    """ViLBERT.forward (vilbert.py:1423-1472) -> ViLBERTForPretraining.forward (:1097-1240) with `visual_target: 0` (or 1: masked-region
    regression, nn.MSELoss over the regions with image_label == 1 / max(their element count, 1), :1139-1148):
    masked_lm_loss = CrossEntropyLoss(ignore_index=-1) over the text stream's prediction scores (HF BertLMPredictionHead, decoder tied
    to the word embeddings); masked_img_loss = sum over the regions with image_label == 1 of KLDivLoss(log_softmax(scores_v),
    cls_prob) / their number (:1150-1157); both `unsqueeze(0)`, keyed "{dataset_name}/{dataset_type}/..." (:1459-1469)."""
    p = prepare_inputs(sample_list)
    seq_t, seq_v, _, _ = vilbert_base(sd, cfg, p["input_ids"], p["image_feature"], p["image_location"], p["token_type_ids"],
                                      p["attention_mask"], p["image_attention_mask"], train)
    eps = cfg["layer_norm_eps"]
    xt = F.gelu(F.linear(seq_t, sd["cls.predictions.transform.dense.weight"], sd["cls.predictions.transform.dense.bias"]))
    xt = layer_norm(xt, sd["cls.predictions.transform.LayerNorm.weight"], sd["cls.predictions.transform.LayerNorm.bias"], eps)
    scores_t = F.linear(xt, sd["bert.embeddings.word_embeddings.weight"], sd["cls.predictions.bias"])
    xv = F.gelu(F.linear(seq_v, sd["cls.imagePredictions.transform.dense.weight"], sd["cls.imagePredictions.transform.dense.bias"]))
    xv = layer_norm(xv, sd["cls.imagePredictions.transform.LayerNorm.weight"], sd["cls.imagePredictions.transform.LayerNorm.bias"], 1e-12)
    scores_v = F.linear(xv, sd["cls.imagePredictions.decoder.weight"], sd["cls.imagePredictions.decoder.bias"])     # :846-858
    image_label = sample_list["image_labels"]
    image_target = torch.as_tensor(sample_list["image_info_0"]["cls_prob"], dtype=torch.float32)                    # :1402-1406
    picked = torch.eq(image_label, 1)
    if cfg.get("visual_target", 0) == 2:                                                                            # :1158-1227
        # `_negative_index` [B, R, K]: the flat region indices the sampling of :1158-1203 produced (test infrastructure hands them in)
        neg = torch.as_tensor(sample_list["_negative_index"]).long()
        B, R, Dv = scores_v.shape
        predict_v = scores_v[picked]                                           # :1205
        neg_v = neg[picked]                                                    # :1206
        flat_target = image_target.reshape(B * R, -1)                          # :1208
        sample_v = torch.cat((image_target[picked].unsqueeze(1), flat_target[neg_v]), dim=1)     # :1210-1212
        score = torch.bmm(sample_v, predict_v.unsqueeze(2)).squeeze(2)         # :1215
        masked_img_loss = F.cross_entropy(score, torch.zeros(score.size(0), dtype=torch.long))  # :1216-1221
    elif cfg.get("visual_target", 0) == 1:                                                                          # :1139-1148
        img_loss = F.mse_loss(scores_v, image_target, reduction="none")
        masked_img_loss = torch.sum(img_loss * picked.unsqueeze(2).float()) / max(torch.sum(picked.unsqueeze(2).expand_as(img_loss)), 1)
    else:
        img_loss = F.kl_div(F.log_softmax(scores_v, dim=2), image_target, reduction="none")                         # :1150-1153
        masked_img_loss = torch.sum(img_loss * picked.unsqueeze(2).float()) / max(torch.sum(picked), 0)             # :1155-1157
    masked_lm_loss = F.cross_entropy(scores_t.view(-1, cfg["vocab_size"]), sample_list["lm_label_ids"].view(-1), ignore_index=-1)
    key = "%s/%s" % (sample_list["dataset_name"], sample_list["dataset_type"])
    return {"losses": {key + "/masked_lm_loss": masked_lm_loss.unsqueeze(0), key + "/masked_img_loss": masked_img_loss.unsqueeze(0)},
            "prediction_scores_t": scores_t, "prediction_scores_v": scores_v}
