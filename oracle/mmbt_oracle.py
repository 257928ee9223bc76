"""CPU oracle for MMF's MMBT path (BASELINE.json configs[0]) — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

fp32 PyTorch restatement of mmf/models/mmbt.py (`ModalEmbeddings` :67-129, `MMBTModel.forward` :176-318,
`MMBTBase.forward` :365-444, `MMBTForClassification.forward` :547-563) over the encoder restated in
oracle/visual_bert_oracle.py (`bert_layer`).  Direct-feature input (`direct_features_input: true`): the modal
encoder is the identity over pre-extracted `[B, N, modal_hidden]` features — the CNN / FRCNN-fc7 encoders
upstream of them are out of scope.

Parity status: PINNED against `tests/golden/mmbt_small64.npz`, produced by running the reference's own
`MMBTBase.forward` + `MMBTModel` + `ModalEmbeddings` + `BertModelJit` (tests/golden/make_golden.py).
Parameter names are the reference's (`model.bert.mmbt.transformer.*`, `model.bert.mmbt.modal_encoder.*`,
`model.classifier.*`); the modal encoder SHARES the text embedding tables and LayerNorm (mmbt.py:78-82).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from oracle.visual_bert_oracle import bert_layer, layer_norm

DEFAULT_CONFIG = dict(
    vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
    max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
    attention_probs_dropout_prob=0.1, modal_hidden_size=2048, num_labels=2, use_modal_start_token=True,
    use_modal_end_token=True, num_segments=2, initializer_range=0.02,
)

T_ = "bert.mmbt.transformer."
M_ = "bert.mmbt.modal_encoder."


def parameter_shapes(cfg):
    """Unique parameters of MMBTForClassification (shared tables listed once, under the transformer)."""
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
        if cfg.get("is_decoder", False):      # BertLayerJit builds a second attention block in decoder mode and never calls it (hf_layers.py:268-292)
            for n in ("query", "key", "value"):
                s[p + "crossattention.self.%s.weight" % n] = (H, H)
                s[p + "crossattention.self.%s.bias" % n] = (H,)
            s[p + "crossattention.output.dense.weight"] = (H, H)
            s[p + "crossattention.output.dense.bias"] = (H,)
            s[p + "crossattention.output.LayerNorm.weight"] = (H,)
            s[p + "crossattention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (I, H)
        s[p + "intermediate.dense.bias"] = (I,)
        s[p + "output.dense.weight"] = (H, I)
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)
    s[T_ + "pooler.dense.weight"] = (H, H)
    s[T_ + "pooler.dense.bias"] = (H,)
    s[M_ + "proj_embeddings.weight"] = (H, cfg["modal_hidden_size"])
    s[M_ + "proj_embeddings.bias"] = (H,)
    s["classifier.0.dense.weight"] = (H, H)
    s["classifier.0.dense.bias"] = (H,)
    s["classifier.0.LayerNorm.weight"] = (H,)
    s["classifier.0.LayerNorm.bias"] = (H,)
    s["classifier.1.weight"] = (cfg["num_labels"], H)
    s["classifier.1.bias"] = (cfg["num_labels"],)
    return s


SHARED = {  # reference state-dict aliases created by mmbt.py:78-82
    M_ + "position_embeddings.weight": T_ + "embeddings.position_embeddings.weight",
    M_ + "token_type_embeddings.weight": T_ + "embeddings.token_type_embeddings.weight",
    M_ + "word_embeddings.weight": T_ + "embeddings.word_embeddings.weight",
    M_ + "LayerNorm.weight": T_ + "embeddings.LayerNorm.weight",
    M_ + "LayerNorm.bias": T_ + "embeddings.LayerNorm.bias",
}


def _vb_view(sd):
    """The encoder restatement in visual_bert_oracle indexes `bert.encoder.layer.N.*`."""
    return {k.replace(T_ + "encoder.", "bert.encoder."): v for k, v in sd.items() if k.startswith(T_ + "encoder.")}


def modal_embeddings(sd, cfg, input_modal, start_token, end_token, token_type_ids, dropout_p=0.0):
    """ModalEmbeddings.forward, mmbt.py:84-129 (identity encoder; position_ids=None -> arange)."""
    e = T_ + "embeddings."
    tok = F.linear(input_modal, sd[M_ + "proj_embeddings.weight"], sd[M_ + "proj_embeddings.bias"])  # :92
    if start_token is not None:
        tok = torch.cat([F.embedding(start_token, sd[e + "word_embeddings.weight"], padding_idx=cfg.get("pad_token_id", 0)).unsqueeze(1), tok], dim=1)  # :95-100
    if end_token is not None:
        tok = torch.cat([tok, F.embedding(end_token, sd[e + "word_embeddings.weight"], padding_idx=cfg.get("pad_token_id", 0)).unsqueeze(1)], dim=1)  # :102-107
    L = tok.size(1)
    position_ids = torch.arange(L, device=tok.device).unsqueeze(0).expand(tok.size(0), L)  # :109-115
    if token_type_ids is None:
        token_type_ids = torch.zeros((tok.size(0), L), dtype=torch.long, device=tok.device)  # :117-122
    emb = tok + F.embedding(position_ids, sd[e + "position_embeddings.weight"]) + F.embedding(
        token_type_ids, sd[e + "token_type_embeddings.weight"])  # :124-126
    emb = layer_norm(emb, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], cfg["layer_norm_eps"])  # :127
    return F.dropout(emb, dropout_p, training=dropout_p > 0)  # :128


def text_embeddings(sd, cfg, input_ids, token_type_ids, dropout_p=0.0):
    """BertEmbeddingsJit.forward, hf_layers.py:108-135."""
    e = T_ + "embeddings."
    T = input_ids.size(1)
    position_ids = torch.arange(T, device=input_ids.device).unsqueeze(0).expand(input_ids.shape)
    emb = (F.embedding(input_ids, sd[e + "word_embeddings.weight"], padding_idx=cfg.get("pad_token_id", 0)) + F.embedding(position_ids, sd[e + "position_embeddings.weight"])
           + F.embedding(token_type_ids, sd[e + "token_type_embeddings.weight"]))
    emb = layer_norm(emb, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], cfg["layer_norm_eps"])
    return F.dropout(emb, dropout_p, training=dropout_p > 0)


def mmbt_model(sd, cfg, input_modal, input_ids, start_tokens, end_tokens, attention_mask, token_type_ids,
               modal_token_type_ids, train=False):
    """MMBTModel.forward, mmbt.py:176-318; `cfg["is_decoder"]`: the causal mask of :260-272 on top of the padding mask."""
    hd = cfg["hidden_dropout_prob"] if train else 0.0
    ad = cfg["attention_probs_dropout_prob"] if train else 0.0
    modal = modal_embeddings(sd, cfg, input_modal, start_tokens, end_tokens, modal_token_type_ids, hd)  # :203-209
    if token_type_ids is None:
        token_type_ids = torch.ones(input_ids.shape, dtype=torch.long, device=input_ids.device)  # :213-216
    txt = text_embeddings(sd, cfg, input_ids, token_type_ids, hd)  # :218-223
    hidden = torch.cat([modal, txt], 1)  # :225
    if attention_mask is None:
        attention_mask = torch.ones(hidden.shape[:-1], device=hidden.device)
    else:
        attention_mask = torch.cat([torch.ones(modal.shape[:-1], dtype=torch.long, device=hidden.device), attention_mask], dim=1)  # :232-238
    if cfg.get("is_decoder", False):  # :260-272: key <= query, times the padding mask, [B, 1, S, S]
        S = hidden.shape[1]
        seq_ids = torch.arange(S, device=hidden.device)
        causal = seq_ids[None, None, :].repeat(hidden.shape[0], S, 1) <= seq_ids[None, :, None]
        ext = (causal[:, None, :, :] * attention_mask[:, None, None, :]).to(hidden.dtype)
    else:
        ext = attention_mask[:, None, None, :].to(hidden.dtype)  # :268-270
    ext = (1.0 - ext) * -10000.0  # :283
    vb = _vb_view(sd)
    for i in range(cfg["num_hidden_layers"]):  # :300-305
        hidden, _ = bert_layer(vb, cfg, i, hidden, ext, hd, ad)
    pooled = torch.tanh(F.linear(hidden[:, 0], sd[T_ + "pooler.dense.weight"], sd[T_ + "pooler.dense.bias"]))  # :308
    return hidden, pooled


def mmbt_base_forward(sd, cfg, sample_list, train=False):
    """MMBTBase.forward + extract_modal_end_token, mmbt.py:346-444."""
    input_modal = sample_list["input_modal"] if "input_modal" in sample_list else sample_list["image_feature_0"]  # :366-370
    input_ids = sample_list["input_ids"]
    input_mask = sample_list["input_mask"]
    start = input_ids[:, 0].clone().detach() if cfg["use_modal_start_token"] else None  # :374-376
    end = None
    if cfg["use_modal_end_token"]:  # :346-363
        gather_index = input_mask.sum(1, keepdim=True) - 1
        end = torch.gather(input_ids, 1, gather_index).squeeze(1).clone().detach()
        B = input_ids.size(0)
        input_ids = torch.cat([input_ids[:, 1:], input_ids[:, -1:]], dim=1)
        input_mask = torch.cat([input_mask[:, 1:], torch.zeros([B, 1], dtype=torch.long, device=input_ids.device)], dim=1)
    if "modal_token_type_ids" in sample_list:
        mtt = sample_list["modal_token_type_ids"]
    else:  # :385-410
        token_value = 0
        seg = sample_list["segment_ids"]
        max_id, min_id = seg.max(), seg.min()
        if max_id == min_id:
            if max_id == 0:
                token_value = 1
        else:
            max_segment = cfg["num_segments"] - 1
            if max_id != max_segment:
                token_value = max_segment
        mtt = torch.full((input_modal.size(0), 1), fill_value=token_value, dtype=torch.long, device=input_modal.device)
    if input_modal.dim() == 2:
        input_modal = input_modal.unsqueeze(dim=1)  # :418-419
    return mmbt_model(sd, cfg, input_modal, input_ids, start, end, input_mask, sample_list["segment_ids"], mtt, train)


def mmbt_forward(sd, cfg, sample_list, train=False):
    """MMBTForClassification.forward, mmbt.py:547-563."""
    seq, pooled = mmbt_base_forward(sd, cfg, sample_list, train)
    hd = cfg["hidden_dropout_prob"] if train else 0.0
    x = F.dropout(pooled, hd, training=hd > 0)
    x = F.gelu(F.linear(x, sd["classifier.0.dense.weight"], sd["classifier.0.dense.bias"]))
    x = layer_norm(x, sd["classifier.0.LayerNorm.weight"], sd["classifier.0.LayerNorm.bias"], cfg["layer_norm_eps"])
    logits = F.linear(x, sd["classifier.1.weight"], sd["classifier.1.bias"])
    return {"scores": logits.contiguous().view(-1, cfg["num_labels"]), "sequence_output": seq, "pooled_output": pooled}


def mmbt_pretraining_forward(sd, cfg, sample_list, train=False):
    """MMBTForPreTraining.forward, mmbt.py:479-523 (masked-LM branch): HF BertPreTrainingHeads.predictions over every position
    (transform = dense -> gelu -> LayerNorm, decoder tied to the word embeddings, :467-476), CrossEntropyLoss(ignore_index=-1)
    over the LAST T positions only (:497-506), keyed "{dataset_name}/{dataset_type}/masked_lm_loss"."""
    seq, pooled = mmbt_base_forward(sd, cfg, sample_list, train)
    x = F.gelu(F.linear(seq, sd["cls.predictions.transform.dense.weight"], sd["cls.predictions.transform.dense.bias"]))
    x = layer_norm(x, sd["cls.predictions.transform.LayerNorm.weight"], sd["cls.predictions.transform.LayerNorm.bias"],
                   cfg["layer_norm_eps"])
    logits = F.linear(x, sd[T_ + "embeddings.word_embeddings.weight"], sd["cls.predictions.bias"])
    lm = sample_list["lm_label_ids"]
    text_scores = logits[:, -(lm.size(1)):].contiguous().view(-1, cfg["vocab_size"])      # :499-503
    loss = F.cross_entropy(text_scores, lm.contiguous().view(-1), ignore_index=-1)
    key = "%s/%s/masked_lm_loss" % (sample_list["dataset_name"], sample_list["dataset_type"])
    return {"logits": logits, "losses": {key: loss}, "sequence_output": seq}


def cross_entropy(scores, targets, **params):
    """CrossEntropyLoss.forward, mmf/modules/losses.py:595-602."""
    return F.cross_entropy(scores, targets, **params)
