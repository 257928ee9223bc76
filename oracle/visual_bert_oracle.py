"""CPU oracle for MMF's VisualBERT cross-modal fusion path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch fp32 restatement of the reference algorithm, written as pure functions over a
state dict whose keys are exactly the reference's parameter names.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; nothing
under `mmf_amd/` does (tests/test_oracle_isolation.py enforces it).

Parity status: PINNED.  `tests/golden/*.npz` hold inputs/outputs produced by running the actual
reference code (`/root/reference/mmf/models/visual_bert.py`, `mmf/modules/embeddings.py`,
`mmf/modules/hf_layers.py`, `mmf/modules/losses.py`, with HF transformers 5.15 for the un-vendored
Bert blocks) in the build container via `tests/golden/make_golden.py`;
`tests/test_oracle_golden.py` checks this file against them (forward, loss and every parameter
gradient).  The reference publishes no golden vectors of its own for this path.

Each function cites the reference lines it follows (paths relative to the reference root; "HF" =
transformers/models/bert/modeling_bert.py, the third-party dependency pinned >=3.4.0,<=4.10.1 by the
reference's requirements.txt:12 and importable here as 5.15.0 with identical block arithmetic).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

DEFAULT_CONFIG = dict(
    # bert-base-uncased (mmf/configs/models/visual_bert/defaults.yaml:3) + VQA2 head
    # (projects/visual_bert/configs/vqa2/defaults.yaml:3-9)
    vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
    intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2,
    layer_norm_eps=1e-12, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
    visual_embedding_dim=2048, num_labels=3129, pooler_strategy="vqa", initializer_range=0.02,
)


# ---------------------------------------------------------------------------------------------
# parameters
# ---------------------------------------------------------------------------------------------
def parameter_shapes(cfg):
    """Reference parameter names and shapes of VisualBERTForClassification (visual_bert.py:284-331),
    i.e. the keys under `model.` in an MMF checkpoint."""
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    s = OrderedDict()
    e = "bert.embeddings."
    s[e + "word_embeddings.weight"] = (cfg["vocab_size"], H)
    s[e + "position_embeddings.weight"] = (cfg["max_position_embeddings"], H)
    s[e + "token_type_embeddings.weight"] = (cfg["type_vocab_size"], H)
    s[e + "LayerNorm.weight"] = (H,)
    s[e + "LayerNorm.bias"] = (H,)
    s[e + "token_type_embeddings_visual.weight"] = (cfg["type_vocab_size"], H)
    s[e + "position_embeddings_visual.weight"] = (cfg["max_position_embeddings"], H)
    s[e + "projection.weight"] = (H, cfg["visual_embedding_dim"])
    s[e + "projection.bias"] = (H,)
    for i in range(cfg["num_hidden_layers"]):
        p = "bert.encoder.layer.%d." % i
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
    s["bert.pooler.dense.weight"] = (H, H)
    s["bert.pooler.dense.bias"] = (H,)
    if cfg.get("bypass_transformer", False):         # visual_bert.py:52-56: one more BertLayer, registered after the pooler
        for k in [k for k in list(s) if k.startswith("bert.encoder.layer.0.")]:
            s["bert.additional_layer." + k[len("bert.encoder.layer.0."):]] = s[k]
    if cfg.get("training_head_type", "classification") == "pretraining":
        # VisualBERTForPretraining (visual_bert.py:160-216): `cls` = HF BertPreTrainingHeads; `cls.predictions.decoder.weight` is the
        # word-embedding table (tie_weights, :227-235) and `cls.predictions.decoder.bias` is `cls.predictions.bias` (HF <= 4.10
        # BertLMPredictionHead), so neither is a separate tensor
        s["cls.predictions.bias"] = (cfg["vocab_size"],)
        s["cls.predictions.transform.dense.weight"] = (H, H)
        s["cls.predictions.transform.dense.bias"] = (H,)
        s["cls.predictions.transform.LayerNorm.weight"] = (H,)
        s["cls.predictions.transform.LayerNorm.bias"] = (H,)
        s["cls.seq_relationship.weight"] = (2, H)
        s["cls.seq_relationship.bias"] = (2,)
        return s
    if cfg.get("training_head_type", "classification") == "nlvr2":
        H = 2 * H   # visual_bert.py:324-325: the head sees the two images' pooled outputs side by side
    s["classifier.0.dense.weight"] = (H, H)
    s["classifier.0.dense.bias"] = (H,)
    s["classifier.0.LayerNorm.weight"] = (H,)
    s["classifier.0.LayerNorm.bias"] = (H,)
    s["classifier.1.weight"] = (cfg["num_labels"], H)
    s["classifier.1.bias"] = (cfg["num_labels"],)
    return s


def init_state_dict(cfg, seed=1234, special_visual_initialize=True):
    """HF `_init_weights` (normal(0, initializer_range) for Linear/Embedding weights, zero biases,
    LayerNorm 1/0) followed by `initialize_visual_from_pretrained` (embeddings.py:321-327,
    visual_bert.py:424-425).  Seeded; not bit-equal to the reference's module-order RNG stream —
    parity never depends on the init, the golden fixtures carry their own weights."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in parameter_shapes(cfg).items():
        if name.endswith("LayerNorm.weight"):
            sd[name] = torch.ones(shape)
        elif name.endswith(".bias"):
            sd[name] = torch.zeros(shape)
        else:
            sd[name] = torch.randn(shape, generator=g) * cfg["initializer_range"]
    if special_visual_initialize:
        e = "bert.embeddings."
        sd[e + "token_type_embeddings_visual.weight"] = sd[e + "token_type_embeddings.weight"].clone()
        sd[e + "position_embeddings_visual.weight"] = sd[e + "position_embeddings.weight"].clone()
    return sd


# ---------------------------------------------------------------------------------------------
# blocks
# ---------------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def embeddings(sd, cfg, input_ids, token_type_ids, visual_embeddings, visual_embeddings_type, dropout_p=0.0,
               image_text_alignment=None):
    """BertVisioLinguisticEmbeddings.forward, embeddings.py:423-459.  `image_text_alignment` None: the else-branch of
    get_position_embeddings_visual (:411-419); a [B, R, A] tensor of text positions (-1 = padding): the masked mean of the
    text position rows plus the visual position row (:375-410)."""
    e = "bert.embeddings."
    T = input_ids.size(1)
    position_ids = torch.arange(T, device=input_ids.device).unsqueeze(0).expand_as(input_ids)  # :332-335
    words = F.embedding(input_ids, sd[e + "word_embeddings.weight"], padding_idx=cfg.get("pad_token_id", 0))  # :339
    pos = F.embedding(position_ids, sd[e + "position_embeddings.weight"])  # :341
    typ = F.embedding(token_type_ids, sd[e + "token_type_embeddings.weight"])  # :342
    text = words + pos + typ  # :343
    out = text
    if visual_embeddings is not None and visual_embeddings_type is not None:
        v = F.linear(visual_embeddings, sd[e + "projection.weight"], sd[e + "projection.bias"])  # :352
        vtyp = F.embedding(visual_embeddings_type, sd[e + "token_type_embeddings_visual.weight"])  # :353-355
        pos_ids_v = torch.zeros(v.shape[:-1], dtype=torch.long, device=v.device)  # :411-415 / :400-404
        vpos = F.embedding(pos_ids_v, sd[e + "position_embeddings_visual.weight"])  # :416-418 / :406-409
        if image_text_alignment is not None:
            am = (image_text_alignment != -1).long()                                  # :379-381
            al = am * image_text_alignment                                            # :383
            ap = F.embedding(al, sd[e + "position_embeddings.weight"]) * am.unsqueeze(-1)   # :387-389
            ap = ap.sum(2)                                                            # :390
            cnt = am.sum(2)                                                           # :393
            cnt = torch.where(cnt == 0, torch.ones_like(cnt), cnt)                    # :394-396
            vpos = ap / cnt.unsqueeze(-1) + vpos                                      # :397-409
        v = v + vpos + vtyp  # :364-368
        out = torch.cat((text, v), dim=1)  # :450-452
    out = layer_norm(out, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], cfg["layer_norm_eps"])  # :457
    return F.dropout(out, dropout_p, training=dropout_p > 0)  # :458


def self_attention(sd, cfg, p, hidden, ext_mask, dropout_p=0.0):
    """BertSelfAttentionJit.forward, hf_layers.py:161-213."""
    A = cfg["num_attention_heads"]
    d = cfg["hidden_size"] // A
    q = F.linear(hidden, sd[p + "query.weight"], sd[p + "query.bias"])  # :169
    k = F.linear(hidden, sd[p + "key.weight"], sd[p + "key.bias"])  # :179
    v = F.linear(hidden, sd[p + "value.weight"], sd[p + "value.bias"])  # :180

    def heads(x):  # transpose_for_scores, :153-159
        return x.view(x.size(0), x.size(1), A, d).permute(0, 2, 1, 3)

    scores = torch.matmul(heads(q), heads(k).transpose(-1, -2))  # :188
    scores = scores / math.sqrt(d)  # :189
    if ext_mask is not None:
        scores = scores + ext_mask  # :193
    probs = F.softmax(scores, dim=-1)  # :196
    probs = F.dropout(probs, dropout_p, training=dropout_p > 0)  # :200
    ctx = torch.matmul(probs, heads(v))  # :206
    ctx = ctx.permute(0, 2, 1, 3).contiguous()  # :208
    return ctx.view(ctx.size(0), ctx.size(1), A * d), probs  # :209-213


def bert_layer(sd, cfg, i, hidden, ext_mask, hidden_dropout=0.0, attn_dropout=0.0):
    """BertLayerJit.forward (hf_layers.py:273-292) -> BertAttentionJit.forward (:233-252) with the HF
    blocks BertSelfOutput (dense, dropout, LayerNorm(x + input)), BertIntermediate (dense, exact-erf
    GELU) and BertOutput (dense, dropout, LayerNorm(x + input))."""
    p = "bert.encoder.layer.%d." % i
    eps = cfg["layer_norm_eps"]
    ctx, probs = self_attention(sd, cfg, p + "attention.self.", hidden, ext_mask, attn_dropout)
    a = F.linear(ctx, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
    a = F.dropout(a, hidden_dropout, training=hidden_dropout > 0)
    a = layer_norm(a + hidden, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], eps)
    h = F.gelu(F.linear(a, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
    o = F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
    o = F.dropout(o, hidden_dropout, training=hidden_dropout > 0)
    o = layer_norm(o + a, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    return o, probs


def visual_bert_base(sd, cfg, input_ids, attention_mask, token_type_ids, visual_embeddings, visual_embeddings_type,
                     train=False, image_text_alignment=None):
    """VisualBERTBase.forward, visual_bert.py:74-157 (bypass_transformer=False)."""
    hd = cfg["hidden_dropout_prob"] if train else 0.0
    ad = cfg["attention_probs_dropout_prob"] if train else 0.0
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)  # :83-84
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)  # :85-86
    ext = attention_mask.unsqueeze(1).unsqueeze(2)  # :94
    ext = ext.to(dtype=sd["bert.embeddings.LayerNorm.weight"].dtype)  # :102-105
    ext = (1.0 - ext) * -10000.0  # :106
    hidden = embeddings(sd, cfg, input_ids, token_type_ids, visual_embeddings, visual_embeddings_type, hd,
                        image_text_alignment)  # :108-114
    if cfg.get("bypass_transformer", False) and visual_embeddings is not None:
        # :116-141 — the text alone goes through the encoder; the visual embeddings join it in ONE `additional_layer`.  The mask
        # slice `extended_attention_mask[:, :, :text_length, :text_length]` (:129-131) acts on a [B, 1, 1, S] tensor: it keeps the
        # broadcast query dimension and the first T keys.
        T = input_ids.size(1)
        text, visual = hidden[:, :T, :], hidden[:, T:, :]
        text_ext = ext[:, :, :T, :T]
        for i in range(cfg["num_hidden_layers"]):
            text, _ = bert_layer(sd, cfg, i, text, text_ext, hd, ad)
        joint = torch.cat((text, visual), dim=1)                                                             # :135
        add = {k.replace("bert.additional_layer.", "bert.encoder.layer.0."): v for k, v in sd.items() if k.startswith("bert.additional_layer.")}
        final, _ = bert_layer(add, cfg, 0, joint, ext, hd, ad)                                               # :136-138
        pooled = torch.tanh(F.linear(final[:, 0], sd["bert.pooler.dense.weight"], sd["bert.pooler.dense.bias"]))   # :139
        return final, pooled, []
    all_hidden = [hidden]
    for i in range(cfg["num_hidden_layers"]):  # BertEncoderJit.forward, hf_layers.py:316-355
        hidden, _ = bert_layer(sd, cfg, i, hidden, ext, hd, ad)
        all_hidden.append(hidden)
    pooled = torch.tanh(F.linear(hidden[:, 0], sd["bert.pooler.dense.weight"], sd["bert.pooler.dense.bias"]))  # :146
    return hidden, pooled, all_hidden


def classification_head(sd, cfg, sequence_output, pooled_output, input_mask, train=False):
    """VisualBERTForClassification.forward, visual_bert.py:369-403."""
    if cfg.get("training_head_type", "classification") == "nlvr2":
        b = pooled_output.size(0)
        pooled_output = torch.cat([pooled_output[: b // 2], pooled_output[b // 2:]], dim=1)  # :369-374, 2B x H -> B x 2H
    if cfg.get("pooler_strategy", "default") == "vqa":
        index = input_mask.sum(1) - 2  # :391
        pooled_output = torch.gather(
            sequence_output, 1,
            index.unsqueeze(-1).unsqueeze(-1).expand(index.size(0), 1, sequence_output.size(-1)))  # :392-398
    hd = cfg["hidden_dropout_prob"] if train else 0.0
    x = F.dropout(pooled_output, hd, training=hd > 0)  # :400
    # BertPredictionHeadTransform: dense -> gelu -> LayerNorm  (visual_bert.py:327-330)
    x = F.gelu(F.linear(x, sd["classifier.0.dense.weight"], sd["classifier.0.dense.bias"]))
    x = layer_norm(x, sd["classifier.0.LayerNorm.weight"], sd["classifier.0.LayerNorm.bias"], cfg["layer_norm_eps"])
    logits = F.linear(x, sd["classifier.1.weight"], sd["classifier.1.bias"])
    return logits.contiguous().view(-1, cfg["num_labels"])  # :402


def prepare_inputs(sample_list):
    """VisualBERT.forward input massaging, visual_bert.py:483-556 + :444-467 (classification head,
    non-nlvr2): returns (input_ids, input_mask, attention_mask, token_type_ids, visual_embeddings,
    visual_embeddings_type)."""
    input_ids = sample_list["input_ids"]
    input_mask = sample_list["input_mask"]
    token_type_ids = sample_list["segment_ids"]
    if "img0" in sample_list and "img1" in sample_list:   # training_head_type == "nlvr2", :490-514: both images, text repeated
        input_ids = torch.cat([input_ids, input_ids])
        input_mask = torch.cat([input_mask, input_mask])
        token_type_ids = torch.cat([token_type_ids, token_type_ids])
        i0, i1 = sample_list["img0"], sample_list["img1"]
        feats = torch.cat([i0["image_feature_0"], i1["image_feature_0"]])
        d0 = (i0.get("image_info_0", None) or {}).get("max_features", None)
        d1 = (i1.get("image_info_0", None) or {}).get("max_features", None)
        image_dim = torch.cat([d0, d1]) if d0 is not None and d1 is not None else None
        info = None
    else:
        feats = sample_list["image_feature_0"]
        image_dim = None
        info = sample_list.get("image_info_0", None)
    if info is not None:
        image_dim = info.get("max_features", None)  # :518-520
    if image_dim is None:
        image_dim = feats.new_full(size=(feats.size(0), 1), fill_value=feats.size(1))  # :525-529
    image_mask = torch.arange(feats.size(-2), device=feats.device).expand(feats.size()[:-1])  # :547-549
    if image_dim.dim() < image_mask.dim():
        image_dim = image_dim.unsqueeze(-1)  # :550-552
    image_mask = (image_mask < image_dim).long()  # :553-554
    vtype = torch.zeros_like(image_mask)  # :447-449
    attention_mask = torch.cat((input_mask, image_mask), dim=-1)  # :450-453
    return input_ids, input_mask, attention_mask, token_type_ids, feats, vtype


def visual_bert_forward(sd, cfg, sample_list, train=False, return_hidden=False):
    """VisualBERT.forward (visual_bert.py:567-601) with training_head_type == "classification"."""
    ids, input_mask, attn_mask, tt, feats, vtype = prepare_inputs(sample_list)
    seq, pooled, all_hidden = visual_bert_base(sd, cfg, ids, attn_mask, tt, feats, vtype, train,
                                               sample_list.get("image_text_alignment", None))   # :585
    scores = classification_head(sd, cfg, seq, pooled, input_mask, train)
    out = {"scores": scores}
    if return_hidden:
        out["sequence_output"] = seq
        out["pooled_output"] = pooled
        out["hidden_states"] = all_hidden
    return out


def pretraining_head(sd, cfg, sequence_output):
    """HF BertPreTrainingHeads.predictions (BertLMPredictionHead: BertPredictionHeadTransform = dense -> gelu -> LayerNorm, then
    the decoder whose weight IS the word-embedding table, visual_bert.py:227-235 `tie_weights`, plus the output-only bias) as
    VisualBERTForPretraining.forward applies it at visual_bert.py:267-269.  The next-sentence head (`cls.seq_relationship`) is
    computed there and never used (no loss term, not returned): left out."""
    x = F.gelu(F.linear(sequence_output, sd["cls.predictions.transform.dense.weight"], sd["cls.predictions.transform.dense.bias"]))
    x = layer_norm(x, sd["cls.predictions.transform.LayerNorm.weight"], sd["cls.predictions.transform.LayerNorm.bias"],
                   cfg["layer_norm_eps"])
    return F.linear(x, sd["bert.embeddings.word_embeddings.weight"], sd["cls.predictions.bias"])


def visual_bert_pretraining_forward(sd, cfg, sample_list, train=False):
    """VisualBERT.forward with training_head_type == "pretraining" (visual_bert.py:567-601): the masked-LM labels `lm_label_ids`
    [B, T] (:539-541) are extended with -1 over the visual positions (:455-465), the loss is CrossEntropyLoss(ignore_index=-1)
    over all B * (T + R) positions (:215, :270-277) and lands in `losses` as "{dataset_name}/{dataset_type}/masked_lm_loss"
    (:588-596).  All labels -1 gives NaN (mean over nothing), which tests/models/test_visual_bert.py:71-98 asserts."""
    ids, input_mask, attn_mask, tt, feats, vtype = prepare_inputs(sample_list)
    seq, pooled, _ = visual_bert_base(sd, cfg, ids, attn_mask, tt, feats, vtype, train)
    logits = pretraining_head(sd, cfg, seq)
    lm = sample_list["lm_label_ids"]
    assert lm.size(-1) == input_mask.size(-1)                      # :456-458
    labels = torch.ones_like(attn_mask) * -1                       # :459
    labels[: lm.size(0), : lm.size(1)] = lm                        # :462-464
    loss = F.cross_entropy(logits.contiguous().view(-1, cfg["vocab_size"]), labels.contiguous().view(-1), ignore_index=-1)
    key = "%s/%s/masked_lm_loss" % (sample_list.get("dataset_name", "random"), sample_list.get("dataset_type", "test"))
    return {"logits": logits, "loss": loss, "losses": {key: loss}, "sequence_output": seq, "pooled_output": pooled}


def logit_bce(scores, targets):
    """LogitBinaryCrossEntropy.forward, mmf/modules/losses.py:246-251."""
    loss = F.binary_cross_entropy_with_logits(scores, targets, reduction="mean")
    return loss * targets.size(1)


def train_step_loss(sd, cfg, sample_list, train=False):
    """forward + loss as BaseModel.__call__ does it (base_model.py:305-337): losses keyed
    "{dataset_type}/{dataset_name}/logit_bce" (losses.py:212-214)."""
    out = visual_bert_forward(sd, cfg, sample_list, train)
    key = "%s/%s/logit_bce" % (sample_list.get("dataset_type", "train"), sample_list.get("dataset_name", "vqa2"))
    out["losses"] = {key: logit_bce(out["scores"], sample_list["targets"])}
    return out


def synthetic_batch(cfg, batch_size, text_len=128, regions=100, seed=1234, full_length=True):
    """The benchmark batch of SURVEY.md §8(d): seeded, full-length text, U[0,1) region features,
    three soft-score targets per row."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, cfg["vocab_size"], (batch_size, text_len), generator=g)
    ids[:, 0] = 101 % cfg["vocab_size"]
    mask = torch.ones(batch_size, text_len, dtype=torch.long)
    if not full_length:
        lens = torch.randint(8, min(25, text_len + 1), (batch_size,), generator=g)
        mask = (torch.arange(text_len)[None, :] < lens[:, None]).long()
    feats = torch.rand(batch_size, regions, cfg["visual_embedding_dim"], generator=g)
    targets = torch.zeros(batch_size, cfg["num_labels"])
    for b in range(batch_size):
        cols = torch.randperm(cfg["num_labels"], generator=g)[:3]
        targets[b, cols] = torch.tensor([1.0, 0.6, 0.3])
    return {
        "input_ids": ids, "input_mask": mask, "segment_ids": torch.zeros_like(ids),
        "image_feature_0": feats,
        "image_info_0": {"max_features": torch.full((batch_size,), regions, dtype=torch.long)},
        "targets": targets, "dataset_name": "vqa2", "dataset_type": "train",
    }
