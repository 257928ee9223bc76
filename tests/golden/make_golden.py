"""Generate tests/golden/visual_bert_*.npz by running the ACTUAL reference implementation
(/root/reference/mmf/models/visual_bert.py + modules/embeddings.py + modules/hf_layers.py +
modules/losses.py, HF transformers for the un-vendored Bert blocks) in the build container.

    python tests/golden/make_golden.py

The reference tree exists only in the build container, so the outputs are committed as small
fixtures; weights are NOT stored — they are regenerated bit-exactly by detweights.state_dict().
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refshim  # noqa: E402

refshim.install()
import detweights  # noqa: E402
from omegaconf import OmegaConf  # noqa: E402  (the shim)
import mmf.models.visual_bert as ref_vb  # noqa: E402
from mmf.modules.losses import LogitBinaryCrossEntropy  # noqa: E402

CASES = {
    # head_dim 16: oracle-only pin
    "tiny": dict(hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64, vocab_size=100,
                 max_position_embeddings=24, visual_embedding_dim=16, num_labels=7, B=3, T=8, R=5, seed=11),
    # head_dim 64, ragged everything: oracle pin + HIP-path pin (tests/test_model_parity_gpu.py)
    "small64": dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=211,
                    max_position_embeddings=40, visual_embedding_dim=72, num_labels=37, B=3, T=12, R=7, seed=12),
}


class SampleList(dict):
    __getattr__ = dict.get


def reference_config(c):
    return OmegaConf.create(dict(
        bert_model_name=None, training_head_type="classification", visual_embedding_dim=c["visual_embedding_dim"],
        special_visual_initialize=True, embedding_strategy="plain", bypass_transformer=False,
        output_attentions=False, output_hidden_states=True, random_initialize=False, freeze_base=False,
        finetune_lr_multiplier=1, pooler_strategy="vqa", zerobias=False,
        hidden_size=c["hidden_size"], num_hidden_layers=c["num_hidden_layers"],
        num_attention_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
        vocab_size=c["vocab_size"], max_position_embeddings=c["max_position_embeddings"], type_vocab_size=2,
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, hidden_act="gelu", layer_norm_eps=1e-12,
        initializer_range=0.02, num_labels=c["num_labels"], losses=[dict(type="logit_bce")], model="visual_bert"))


def make_inputs(c):
    seed, B, T, R = c["seed"], c["B"], c["T"], c["R"]
    ids = (detweights.uniform(B * T, seed + 100) * c["vocab_size"]).astype(np.int64).reshape(B, T)
    mask = np.ones((B, T), dtype=np.int64)
    mask[1, T // 2:] = 0           # a padded question
    mask[2, T - 2:] = 0
    ids[mask == 0] = 0             # [PAD] = pad_token_id: HF's word_embeddings has padding_idx=0 (no gradient for that row)
    seg = (detweights.uniform(B * T, seed + 101) > 0.7).astype(np.int64).reshape(B, T)
    feats = detweights.uniform(B * R * c["visual_embedding_dim"], seed + 102).astype(np.float32).reshape(B, R, -1)
    max_features = np.array([R, R - 2, R - 1], dtype=np.int64)[:B]
    targets = np.zeros((B, c["num_labels"]), dtype=np.float32)
    for b in range(B):
        targets[b, (3 * b + 1) % c["num_labels"]] = 1.0
        targets[b, (5 * b + 2) % c["num_labels"]] = 0.6
    return dict(input_ids=ids, input_mask=mask, segment_ids=seg, image_feature_0=feats, max_features=max_features,
                targets=targets)


def make_alignment(c):
    """`image_text_alignment` [B, R, A]: text positions each region is aligned to, -1 = padding (embeddings.py:375-378); one region
    has no aligned word at all (the divide-by-zero guard, :394-396), one is aligned to the same word twice."""
    B, T, R, A = c["B"], c["T"], c["R"], 3
    al = (detweights.uniform(B * R * A, c["seed"] + 103) * (T + 3)).astype(np.int64).reshape(B, R, A) - 3
    al[al < 0] = -1
    al[0, 1, :] = -1
    al[1, 2, :] = [4, 4, -1]
    return al


def main(align=False):
    for name, c in ({"align64": CASES["small64"]} if align else CASES).items():
        cfg = reference_config(c)
        model = ref_vb.VisualBERT(cfg)
        model.build()
        model.eval()  # dropout off: parity is defined in eval mode (SURVEY.md §7 hard parts)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("position_ids")}
        sd = detweights.state_dict(shapes, c["seed"])
        missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not unexpected and all(k.endswith("position_ids") for k in missing), (missing, unexpected)
        inp = make_inputs(c)
        sl = SampleList(
            input_ids=torch.from_numpy(inp["input_ids"]), input_mask=torch.from_numpy(inp["input_mask"]),
            segment_ids=torch.from_numpy(inp["segment_ids"]), image_feature_0=torch.from_numpy(inp["image_feature_0"]),
            image_info_0=SampleList(max_features=torch.from_numpy(inp["max_features"])),
            targets=torch.from_numpy(inp["targets"]), dataset_name="vqa2", dataset_type="train")
        if align:
            inp["image_text_alignment"] = make_alignment(c)
            sl["image_text_alignment"] = torch.from_numpy(inp["image_text_alignment"])   # read at visual_bert.py:585
        out = model.forward(sl)
        loss = LogitBinaryCrossEntropy()(sl, out)
        loss.backward()
        rec = {"in_" + k: v for k, v in inp.items()}
        rec["scores"] = out["scores"].detach().numpy()
        rec["sequence_output"] = out["sequence_output"].detach().numpy()
        rec["pooled_output"] = out["pooled_output"].detach().numpy()
        rec["loss"] = np.array(loss.item(), dtype=np.float64)
        names, norms, sums = [], [], []
        for k, p in model.named_parameters():
            g = p.grad
            names.append(k)
            norms.append(0.0 if g is None else float(g.double().norm()))
            sums.append(0.0 if g is None else float(g.double().sum()))
            # full gradients of the small tensors (biases, LayerNorms, type tables)
            if g is not None and g.numel() <= 4096:
                rec["grad::" + k] = g.numpy()
        rec["grad_names"] = np.array(names)
        rec["grad_norms"] = np.array(norms)
        rec["grad_sums"] = np.array(sums)
        rec["param_names"] = np.array(list(shapes.keys()))
        rec["param_shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
        rec["case"] = np.array(repr(c))
        path = os.path.join(HERE, "visual_bert_%s.npz" % name)
        np.savez_compressed(path, **rec)
        print(name, "loss", loss.item(), "scores[0,:4]", rec["scores"][0, :4], "->", path, os.path.getsize(path), "bytes")


MMBT_CASES = {
    "mmbt_small64": dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=211,
                         max_position_embeddings=40, modal_hidden_size=72, num_labels=2, B=4, T=12, N=7, seed=21),
    # MMBT as a decoder (mmbt.py:244-266: the padding mask times a causal mask over the modal + text positions; BertLayerJit then also owns an
    # unused `crossattention` block, hf_layers.py:268-271)
    "mmbt_decoder64": dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=211,
                           max_position_embeddings=40, modal_hidden_size=72, num_labels=2, B=4, T=12, N=7, seed=23, is_decoder=True),
}


def make_mmbt():
    """MMBT (BASELINE configs[0]) through the reference's own MMBTBase.forward / MMBTModel / ModalEmbeddings /
    BertModelJit and MMBTForClassification.forward, direct-feature input (identity modal encoder)."""
    from torch import nn
    from transformers import BertConfig
    M = refshim.ref_import("mmf.models.mmbt")
    from mmf.modules.hf_layers import BertModelJit
    from mmf.modules.losses import CrossEntropyLoss
    from transformers.models.bert.modeling_bert import BertPredictionHeadTransform

    for name, c in MMBT_CASES.items():
        bcfg = BertConfig(hidden_size=c["hidden_size"], num_hidden_layers=c["num_hidden_layers"],
                          num_attention_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
                          vocab_size=c["vocab_size"], max_position_embeddings=c["max_position_embeddings"], type_vocab_size=2,
                          hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12, is_decoder=bool(c.get("is_decoder", False)))
        mcfg = M.MMBTConfig(bcfg, num_labels=c["num_labels"], modal_hidden_size=c["modal_hidden_size"])

        class Holder(nn.Module):
            pass

        class RefMMBT(nn.Module):   # module tree of MMBTForClassification: bert.mmbt.*, classifier.*
            def __init__(self):
                super().__init__()
                self.bert = Holder()
                self.bert.mmbt = M.MMBTModel(mcfg, BertModelJit(bcfg), nn.Identity())
                self.classifier = nn.Sequential(BertPredictionHeadTransform(bcfg), nn.Linear(c["hidden_size"], c["num_labels"]))

        ref = RefMMBT().eval()
        uniq = {k: tuple(v.shape) for k, v in ref.state_dict().items()
                if not k.endswith("position_ids") and not k.endswith("token_type_ids") and "modal_encoder.position_embeddings" not in k
                and "modal_encoder.token_type_embeddings" not in k and "modal_encoder.word_embeddings" not in k
                and "modal_encoder.LayerNorm" not in k}
        sd = detweights.state_dict(uniq, c["seed"])
        missing, unexpected = ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not unexpected, unexpected
        # MMBTBase.forward needs only these attributes of `self`
        base = Holder()
        base._is_direct_features_input = True
        base.use_modal_start_token = True
        base.use_modal_end_token = True
        base.num_max_segment = 2
        base.mmbt = ref.bert.mmbt
        base.extract_modal_end_token = lambda sl: M.MMBTBase.extract_modal_end_token(base, sl)
        head = Holder()
        head.bert = lambda sl: M.MMBTBase.forward(base, sl)
        head.output_hidden_states = False
        head.output_attentions = False
        head.fused_feature_only = False
        head.dropout = nn.Dropout(0.1).eval()
        head.classifier = ref.classifier
        head.num_labels = c["num_labels"]

        B, T, N, seed = c["B"], c["T"], c["N"], c["seed"]
        ids = (detweights.uniform(B * T, seed + 100) * c["vocab_size"]).astype(np.int64).reshape(B, T)
        mask = np.ones((B, T), dtype=np.int64)
        mask[1, T // 2:] = 0
        mask[2, T - 3:] = 0
        ids[mask == 0] = 0   # [PAD]
        seg = np.zeros((B, T), dtype=np.int64)
        feats = detweights.uniform(B * N * c["modal_hidden_size"], seed + 102).astype(np.float32).reshape(B, N, -1)
        targets = (detweights.uniform(B, seed + 103) > 0.5).astype(np.int64)
        sl = SampleList(input_ids=torch.from_numpy(ids.copy()), input_mask=torch.from_numpy(mask.copy()),
                        segment_ids=torch.from_numpy(seg), image_feature_0=torch.from_numpy(feats),
                        targets=torch.from_numpy(targets), dataset_name="hateful_memes", dataset_type="train")
        seq_holder = {}
        orig_forward = ref.bert.mmbt.forward

        def spy(*a, **k):
            out = orig_forward(*a, **k)
            seq_holder["seq"], seq_holder["pooled"] = out[0], out[1]
            return out

        ref.bert.mmbt.forward = spy
        out = M.MMBTForClassification.forward(head, sl)
        loss = CrossEntropyLoss()(sl, out)
        loss.backward()
        rec = {"in_input_ids": ids, "in_input_mask": mask, "in_segment_ids": seg, "in_image_feature_0": feats, "in_targets": targets}
        rec["scores"] = out["scores"].detach().numpy()
        rec["sequence_output"] = seq_holder["seq"].detach().numpy()
        rec["pooled_output"] = seq_holder["pooled"].detach().numpy()
        rec["loss"] = np.array(loss.item(), dtype=np.float64)
        names, norms, sums = [], [], []
        seen = set()
        for k, p in ref.named_parameters():   # named_parameters lists shared parameters once
            if id(p) in seen:
                continue
            seen.add(id(p))
            g = p.grad
            names.append("model." + k)
            norms.append(0.0 if g is None else float(g.double().norm()))
            sums.append(0.0 if g is None else float(g.double().sum()))
            if g is not None and g.numel() <= 4096:
                rec["grad::model." + k] = g.numpy()
        rec["grad_names"] = np.array(names)
        rec["grad_norms"] = np.array(norms)
        rec["grad_sums"] = np.array(sums)
        rec["param_names"] = np.array(["model." + k for k in uniq.keys()])
        rec["param_shapes"] = np.array([",".join(map(str, s)) for s in uniq.values()])
        rec["state_dict_keys"] = np.array(["model." + k for k in ref.state_dict().keys()])
        rec["case"] = np.array(repr(c))
        path = os.path.join(HERE, "%s.npz" % name)
        np.savez_compressed(path, **rec)
        print(name, "loss", loss.item(), "scores", rec["scores"][0], "->", path, os.path.getsize(path), "bytes")

def make_mmbt_pretraining():
    """MMBTForPreTraining.forward (mmbt.py:479-523) called on a holder that carries the reference's own parts (MMBTBase.forward over
    MMBTModel / BertModelJit, HF BertPreTrainingHeads with the decoder tied the pinned-transformers way): masked-LM loss over the
    text positions, logits over all positions, every gradient."""
    from torch import nn
    from transformers import BertConfig
    from transformers.models.bert.modeling_bert import BertPreTrainingHeads
    M = refshim.ref_import("mmf.models.mmbt")
    from mmf.modules.hf_layers import BertModelJit
    c = dict(MMBT_CASES["mmbt_small64"], seed=81)
    bcfg = BertConfig(hidden_size=c["hidden_size"], num_hidden_layers=c["num_hidden_layers"],
                      num_attention_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
                      vocab_size=c["vocab_size"], max_position_embeddings=c["max_position_embeddings"], type_vocab_size=2,
                      hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12)
    mcfg = M.MMBTConfig(bcfg, num_labels=c["num_labels"], modal_hidden_size=c["modal_hidden_size"])

    class Holder(nn.Module):
        pass

    class RefMMBT(nn.Module):   # module tree of MMBTForPreTraining: bert.mmbt.*, cls.*
        def __init__(self):
            super().__init__()
            self.bert = Holder()
            self.bert.mmbt = M.MMBTModel(mcfg, BertModelJit(bcfg), nn.Identity())
            self.cls = BertPreTrainingHeads(bcfg)
            # mmbt.py:467-476 tie_weights (+ transformers<=4.10 BertLMPredictionHead: decoder.bias IS predictions.bias)
            self.cls.predictions.decoder.weight = self.bert.mmbt.transformer.embeddings.word_embeddings.weight
            self.cls.predictions.decoder.bias = self.cls.predictions.bias

    ref = RefMMBT().eval()
    uniq = {k: tuple(v.shape) for k, v in ref.state_dict().items()
            if not k.endswith("position_ids") and not k.endswith("token_type_ids") and "modal_encoder.position_embeddings" not in k
            and "modal_encoder.token_type_embeddings" not in k and "modal_encoder.word_embeddings" not in k
            and "modal_encoder.LayerNorm" not in k and not k.startswith("cls.predictions.decoder.")}
    sd = detweights.state_dict(uniq, c["seed"])
    missing, unexpected = ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    base = Holder()
    base._is_direct_features_input = True
    base.use_modal_start_token = True
    base.use_modal_end_token = True
    base.num_max_segment = 2
    base.mmbt = ref.bert.mmbt
    base.extract_modal_end_token = lambda sl: M.MMBTBase.extract_modal_end_token(base, sl)
    head = Holder()
    head.bert = lambda sl: M.MMBTBase.forward(base, sl)
    head.cls = ref.cls
    head.encoder_config = bcfg
    head.loss_fct = nn.CrossEntropyLoss(ignore_index=-1)

    B, T, N, seed = c["B"], c["T"], c["N"], c["seed"]
    ids = (detweights.uniform(B * T, seed + 100) * c["vocab_size"]).astype(np.int64).reshape(B, T)
    mask = np.ones((B, T), dtype=np.int64)
    mask[1, T // 2:] = 0
    mask[2, T - 3:] = 0
    ids[mask == 0] = 0
    seg = np.zeros((B, T), dtype=np.int64)
    feats = detweights.uniform(B * N * c["modal_hidden_size"], seed + 102).astype(np.float32).reshape(B, N, -1)
    pick = (detweights.uniform(B * T, seed + 301).reshape(B, T) < 0.3) & (mask == 1)
    pick[:, 2] = True
    lm = np.where(pick, (detweights.uniform(B * T, seed + 302) * c["vocab_size"]).astype(np.int64).reshape(B, T), -1)
    sl = SampleList(input_ids=torch.from_numpy(ids.copy()), input_mask=torch.from_numpy(mask.copy()),
                    segment_ids=torch.from_numpy(seg), image_feature_0=torch.from_numpy(feats),
                    lm_label_ids=torch.from_numpy(lm), dataset_name="hateful_memes", dataset_type="train")
    out = M.MMBTForPreTraining.forward(head, sl)
    (key, loss), = out["losses"].items()
    loss.backward()
    rec = {"in_input_ids": ids, "in_input_mask": mask, "in_segment_ids": seg, "in_image_feature_0": feats, "in_lm_label_ids": lm}
    rec["logits"] = out["logits"].detach().numpy()
    rec["loss"] = np.array(loss.item(), dtype=np.float64)
    rec["loss_key"] = np.array(key)
    names, norms, sums = [], [], []
    seen = set()
    for k, p in ref.named_parameters():
        if id(p) in seen:
            continue
        seen.add(id(p))
        g = p.grad
        names.append("model." + k)
        norms.append(0.0 if g is None else float(g.double().norm()))
        sums.append(0.0 if g is None else float(g.double().sum()))
        if g is not None and (g.numel() <= 4096 or k.endswith("word_embeddings.weight")):
            rec["grad::model." + k] = g.numpy()
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array(norms)
    rec["grad_sums"] = np.array(sums)
    rec["param_names"] = np.array(["model." + k for k in uniq.keys()])
    rec["param_shapes"] = np.array([",".join(map(str, s)) for s in uniq.values()])
    rec["state_dict_keys"] = np.array(["model." + k for k in ref.state_dict().keys()])
    rec["case"] = np.array(repr(c))
    path = os.path.join(HERE, "mmbt_pretraining.npz")
    np.savez_compressed(path, **rec)
    print("mmbt_pretraining loss", loss.item(), key, "->", path, os.path.getsize(path), "bytes")


MMFT_CASES = {
    "mmft_small64": dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=211,
                         max_position_embeddings=40, embedding_dim=72, num_labels=5, B=3, T=12, R=7, seed=31),
}


def make_mmft():
    """MMF Transformer (mmft) through the reference's own MMFTransformer.forward / preprocess_sample,
    BaseTransformerBackend.forward + HuggingfaceBackend.generate_* , HuggingfaceEmbeddings, BertModelJit encoder and the
    MLP head.  The pieces that need the network in the reference (`AutoConfig/BertModelJit.from_pretrained`) are replaced by
    a randomly-initialised BertModelJit of the same class; the encoder is called the way the reference's scripted branch
    calls it (huggingface.py:227) — its eager branch (:229-231) passes `[None]*L` in the `encoder_hidden_states` slot of
    this tree's BertEncoderJit and cannot run."""
    import types
    from torch import nn
    from transformers import BertConfig
    M = refshim.ref_import("mmf.models.mmf_transformer")
    HB = refshim.ref_import("mmf.models.transformers.backends.huggingface")
    TB = refshim.ref_import("mmf.models.transformers.base")
    from mmf.models.transformers.heads.mlp import MLP
    from mmf.modules.hf_layers import BertModelJit
    from mmf.modules.losses import CrossEntropyLoss

    for name, c in MMFT_CASES.items():
        torch.manual_seed(c["seed"])
        bcfg = BertConfig(hidden_size=c["hidden_size"], num_hidden_layers=c["num_hidden_layers"],
                          num_attention_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
                          vocab_size=c["vocab_size"], max_position_embeddings=c["max_position_embeddings"], type_vocab_size=2,
                          hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12, pad_token_id=0)
        mcfg = OmegaConf.create(dict(
            model="mmft", transformer_base="bert-base-uncased", num_labels=c["num_labels"], initializer_range=0.02,
            initializer_mean=0.0, token_noise_std=0.01, token_noise_mean=0.0, layer_norm_weight_fill=1.0, random_initialize=False,
            heads=[dict(type="mlp", hidden_size=c["hidden_size"], num_labels=c["num_labels"])],
            modalities=[
                dict(type="text", key="text", position_dim=c["max_position_embeddings"], segment_id=0,
                     embedding_dim=c["hidden_size"], layer_norm_eps=1e-12, hidden_dropout_prob=0.1),
                dict(type="image", key="image", embedding_dim=c["embedding_dim"], position_dim=c["max_position_embeddings"],
                     segment_id=1, layer_norm_eps=1e-12, hidden_dropout_prob=0.1),
            ]))

        class RefBackend(nn.Module):
            def __init__(self):
                super().__init__()
                self.config = mcfg
                self.transformer_config = bcfg
                self.transformer = BertModelJit(bcfg)
                self.embeddings = HB.HuggingfaceEmbeddings(mcfg, bcfg, self.transformer)

            generate_embeddings = HB.HuggingfaceBackend.generate_embeddings
            generate_attention_mask = HB.HuggingfaceBackend.generate_attention_mask
            forward = TB.BaseTransformerBackend.forward

            def generate_encoded_layers(self, embedding, attention_mask):
                encoded_layers = self.transformer.encoder(embedding, attention_mask)   # huggingface.py:227
                return encoded_layers[-1], encoded_layers[0]                            # :232

        class RefMMFT(nn.Module):   # module tree of MMFTransformer: backend.*, encoders.*, heads.*
            def __init__(self):
                super().__init__()
                self.config = mcfg
                self.backend = RefBackend()
                self.encoders = nn.ModuleDict({"text": nn.Identity(), "image": nn.Identity()})
                self.heads = nn.ModuleList([MLP(mcfg.heads[0])])
                self.modality_keys = ["text", "image"]
                self.modality_type = ["text", "image"]
                self.modality_segments = [0, 1]

        for fn in ("forward", "preprocess_sample", "_infer_input_ids", "_check_keys_for_modality", "_infer_position_ids",
                   "_infer_masks", "_infer_segment_ids", "_infer_itm_labels", "_infer_mlm_labels", "postprocess_output"):
            setattr(RefMMFT, fn, getattr(M.MMFTransformer, fn))
        ref = RefMMFT().eval()
        full = {k: tuple(v.shape) for k, v in ref.state_dict().items() if not k.endswith("position_ids") and not k.endswith("embeddings.token_type_ids")}
        alias = {"backend.embeddings.token_embeddings.0.weight": "backend.transformer.embeddings.word_embeddings.weight",
                 "backend.embeddings.layer_norms.0.weight": "backend.transformer.embeddings.LayerNorm.weight",
                 "backend.embeddings.layer_norms.0.bias": "backend.transformer.embeddings.LayerNorm.bias"}
        uniq = {k: v for k, v in full.items() if k not in alias}
        sd = detweights.state_dict(uniq, c["seed"])
        missing, unexpected = ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not unexpected, unexpected
        assert ref.backend.embeddings.token_embeddings[0].weight is ref.backend.transformer.embeddings.word_embeddings.weight

        B, T, R, seed = c["B"], c["T"], c["R"], c["seed"]
        ids = (detweights.uniform(B * T, seed + 100) * c["vocab_size"]).astype(np.int64).reshape(B, T)
        mask = np.ones((B, T), dtype=np.int64)
        mask[1, T // 2:] = 0
        mask[2, T - 3:] = 0
        ids[mask == 0] = 0   # [PAD]
        seg = np.zeros((B, T), dtype=np.int64)
        feats = detweights.uniform(B * R * c["embedding_dim"], seed + 102).astype(np.float32).reshape(B, R, -1)
        image_mask = np.ones((B, R), dtype=np.int64)
        image_mask[1, R - 2:] = 0
        targets = (detweights.uniform(B, seed + 103) * c["num_labels"]).astype(np.int64)
        sl = SampleList(input_ids=torch.from_numpy(ids), input_mask=torch.from_numpy(mask), segment_ids=torch.from_numpy(seg),
                        image=torch.from_numpy(feats), image_mask=torch.from_numpy(image_mask), targets=torch.from_numpy(targets),
                        dataset_name="hateful_memes", dataset_type="train")
        holder = {}
        orig = ref.backend.forward

        def spy(*a, **k):
            out = orig(*a, **k)
            holder["seq"] = out[0]
            return out

        ref.backend.forward = spy
        out = ref(sl)
        loss = CrossEntropyLoss()(sl, out)
        loss.backward()
        rec = {"in_input_ids": ids, "in_input_mask": mask, "in_segment_ids": seg, "in_image": feats, "in_image_mask": image_mask,
               "in_targets": targets}
        rec["scores"] = out["scores"].detach().numpy()
        rec["sequence_output"] = holder["seq"].detach().numpy()
        rec["loss"] = np.array(loss.item(), dtype=np.float64)
        names, norms, sums = [], [], []
        for k, p in ref.named_parameters():
            g = p.grad
            names.append(k)
            norms.append(0.0 if g is None else float(g.double().norm()))
            sums.append(0.0 if g is None else float(g.double().sum()))
            if g is not None and g.numel() <= 4096:
                rec["grad::" + k] = g.numpy()
        rec["grad_names"] = np.array(names)
        rec["grad_norms"] = np.array(norms)
        rec["grad_sums"] = np.array(sums)
        rec["param_names"] = np.array(list(uniq.keys()))
        rec["param_shapes"] = np.array([",".join(map(str, s)) for s in uniq.values()])
        rec["state_dict_keys"] = np.array(list(ref.state_dict().keys()))
        rec["case"] = np.array(repr(c))
        path = os.path.join(HERE, "%s.npz" % name)
        np.savez_compressed(path, **rec)
        print(name, "loss", loss.item(), "scores", rec["scores"][0], "->", path, os.path.getsize(path), "bytes")


VILBERT_NLVR2 = dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2,
                     v_hidden_size=256, v_num_attention_heads=2, v_intermediate_size=192, v_num_hidden_layers=2,
                     bi_hidden_size=256, bi_num_attention_heads=2, bi_intermediate_size=256,
                     v_biattention_id=[0], t_biattention_id=[1], vocab_size=211, max_position_embeddings=40,
                     v_feature_size=72, num_labels=2, B=3, T=12, R=7, seed=47, nlvr2=True)

VILBERT_CASES = {
    # text stream d=64, visual + co-attention streams d=128 (as in the real config: 768/12, 1024/8, 1024/8)
    "vilbert_small": dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=4,
                          v_hidden_size=256, v_num_attention_heads=2, v_intermediate_size=192, v_num_hidden_layers=3,
                          bi_hidden_size=256, bi_num_attention_heads=2, bi_intermediate_size=256,
                          v_biattention_id=[0, 1], t_biattention_id=[1, 2], vocab_size=211, max_position_embeddings=40,
                          v_feature_size=72, num_labels=11, B=3, T=12, R=7, seed=42),   # seed picked for bf16 conditioning, see test_vilbert_gpu.py
    # the same network with `dynamic_attention: true`: visual self-attention queries / keys gated by 1 + sigmoid(Linear(masked mean
    # of the text stream)) (vilbert.py:174-176, 199-212)
    "vilbert_dyn": dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=4,
                        v_hidden_size=256, v_num_attention_heads=2, v_intermediate_size=192, v_num_hidden_layers=3,
                        bi_hidden_size=256, bi_num_attention_heads=2, bi_intermediate_size=256,
                        v_biattention_id=[0, 1], t_biattention_id=[1, 2], vocab_size=211, max_position_embeddings=40,
                        v_feature_size=72, num_labels=11, B=3, T=12, R=7, seed=42, dynamic_attention=True),
    # `fixed_t_layer: 2, fixed_v_layer: 1` (vilbert.py:625-666): the first text / visual layer runs without gradient; the reference's loop
    # sets t_start = fixed_t_layer at that first layer, so text layer 1 is SKIPPED (never executed) — recorded as the reference behaves
    "vilbert_fixed": dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=4,
                          v_hidden_size=256, v_num_attention_heads=2, v_intermediate_size=192, v_num_hidden_layers=3,
                          bi_hidden_size=256, bi_num_attention_heads=2, bi_intermediate_size=256,
                          v_biattention_id=[1, 2], t_biattention_id=[2, 3], vocab_size=211, max_position_embeddings=40,
                          v_feature_size=72, num_labels=11, B=3, T=12, R=7, seed=44, fixed_t_layer=2, fixed_v_layer=1),
    # `in_batch_pairs: true` (vilbert.py:678-710): at the first connection point the batch becomes every text against every image (B^2
    # pairs); the scores are [B^2, num_labels], the targets of this case too
    "vilbert_pairs": dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=4,
                          v_hidden_size=256, v_num_attention_heads=2, v_intermediate_size=192, v_num_hidden_layers=3,
                          bi_hidden_size=256, bi_num_attention_heads=2, bi_intermediate_size=256,
                          v_biattention_id=[0, 1], t_biattention_id=[1, 2], vocab_size=211, max_position_embeddings=40,
                          v_feature_size=72, num_labels=11, B=3, T=12, R=7, seed=45, in_batch_pairs=True),
    # `fast_mode: true` (vilbert.py:712-723): ONE text against B images, the text stream expanded at the first connection point.  The batch
    # sizes differ, so this case calls ViLBERTForClassification.forward directly (a SampleList cannot hold it)
    "vilbert_fast": dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=4,
                         v_hidden_size=256, v_num_attention_heads=2, v_intermediate_size=192, v_num_hidden_layers=3,
                         bi_hidden_size=256, bi_num_attention_heads=2, bi_intermediate_size=256,
                         v_biattention_id=[0, 1], t_biattention_id=[1, 2], vocab_size=211, max_position_embeddings=40,
                         v_feature_size=72, num_labels=11, B=3, T=12, R=7, seed=46, fast_mode=True),
}


def vilbert_reference_config(c):
    return OmegaConf.create(dict(
        bert_model_name=None, training_head_type="classification", visual_embedding_dim=c["v_feature_size"],
        special_visual_initialize=True, hard_cap_seq_len=None, cut_first="text", embedding_strategy="plain",
        bypass_transformer=False, output_attentions=False, output_hidden_states=False, text_only=False, random_initialize=False,
        freeze_base=False, finetune_lr_multiplier=1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12, hidden_act="gelu",
        hidden_dropout_prob=0.1, hidden_size=c["hidden_size"], initializer_range=0.02, intermediate_size=c["intermediate_size"],
        max_position_embeddings=c["max_position_embeddings"], num_attention_heads=c["num_attention_heads"],
        num_hidden_layers=c["num_hidden_layers"], type_vocab_size=2, vocab_size=c["vocab_size"],
        v_feature_size=c["v_feature_size"], v_target_size=1601, v_hidden_size=c["v_hidden_size"],
        v_num_hidden_layers=c["v_num_hidden_layers"], v_num_attention_heads=c["v_num_attention_heads"],
        v_intermediate_size=c["v_intermediate_size"], bi_hidden_size=c["bi_hidden_size"],
        bi_num_attention_heads=c["bi_num_attention_heads"], bi_intermediate_size=c["bi_intermediate_size"], bi_attention_type=1,
        v_attention_probs_dropout_prob=0.1, v_hidden_act="gelu", v_hidden_dropout_prob=0.1, v_initializer_range=0.02,
        v_biattention_id=c["v_biattention_id"], t_biattention_id=c["t_biattention_id"], pooling_method="mul", fusion_method="mul",
        fast_mode=bool(c.get("fast_mode", False)), with_coattention=True, dynamic_attention=bool(c.get("dynamic_attention", False)), in_batch_pairs=bool(c.get("in_batch_pairs", False)),
        task_specific_tokens=False,
        fixed_v_layer=int(c.get("fixed_v_layer", 0)), fixed_t_layer=int(c.get("fixed_t_layer", 0)), visualization=False, visual_target=0, objective=0,
        num_negative=128, model="vilbert",
        num_labels=c["num_labels"], losses=[dict(type="logit_bce")]))


def make_vilbert(only=None):
    """ViLBERT (two streams + co-attention, classification head) through the reference's own ViLBERT.forward /
    get_image_and_text_features, ViLBERTForClassification.forward, ViLBERTBase, BertEncoder, BertConnectionLayer,
    BertBiAttention, BertImageLayer, ... (mmf/models/vilbert.py).  Only `ViLBERTBase.from_pretrained` (network) is replaced
    by constructing `ViLBERTBase(config)` directly."""
    from copy import deepcopy
    from torch import nn
    from transformers import BertConfig
    M = refshim.ref_import("mmf.models.vilbert")
    from transformers.models.bert.modeling_bert import BertPredictionHeadTransform
    # replace_with_jit() (vilbert.py:920) monkey-patches HF's own BertSelfAttention / BertLayer / BertEncoder for
    # TorchScript; ViLBERT never calls those classes (it defines its own attention) and the patch targets methods that the
    # installed transformers no longer has.
    M.replace_with_jit = lambda: None

    cases = dict(VILBERT_CASES)
    cases["vilbert_nlvr2"] = VILBERT_NLVR2
    if only is not None:
        cases = {k: v for k, v in cases.items() if k in only}
    for name, c in cases.items():
        nlvr2 = bool(c.get("nlvr2", False))
        torch.manual_seed(c["seed"])
        cfg = vilbert_reference_config(c)
        if nlvr2:
            cfg["training_head_type"] = "nlvr2"
        bcfg = BertConfig.from_dict(OmegaConf.to_container(cfg))

        class RefCls(nn.Module):   # module tree of ViLBERTForClassification (vilbert.py:1243-1270)
            def __init__(self):
                super().__init__()
                self.config = cfg
                self.bert = M.ViLBERTBase(bcfg)
                self.training_head_type = "nlvr2" if nlvr2 else "classification"
                self.num_labels = c["num_labels"]
                self.fusion_method = "mul"
                self.dropout = nn.Dropout(0.1)
                ccfg = deepcopy(bcfg)
                ccfg.hidden_size = c["bi_hidden_size"] * (2 if nlvr2 else 1)      # vilbert.py:1262-1265
                self.classifier = nn.Sequential(BertPredictionHeadTransform(ccfg), nn.Linear(ccfg.hidden_size, c["num_labels"]))

            forward = M.ViLBERTForClassification.forward

        class RefViLBERT(nn.Module):   # the registered BaseModel: `model.*`
            def __init__(self):
                super().__init__()
                self.config = cfg
                self.model = RefCls()

            get_image_and_text_features = M.ViLBERT.get_image_and_text_features
            forward = M.ViLBERT.forward

        ref = RefViLBERT().eval()
        shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()
                  if not k.endswith("position_ids") and not k.endswith("embeddings.token_type_ids")}
        sd = detweights.state_dict(shapes, c["seed"])
        missing, unexpected = ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not unexpected, unexpected
        B, T, R, seed = c["B"], c["T"], c["R"], c["seed"]
        ids = (detweights.uniform(B * T, seed + 100) * c["vocab_size"]).astype(np.int64).reshape(B, T)
        mask = np.ones((B, T), dtype=np.int64)
        mask[1, T // 2:] = 0
        mask[2, T - 3:] = 0
        ids[mask == 0] = 0   # [PAD]
        seg = np.zeros((B, T), dtype=np.int64)
        # zero-mean region features: identical-looking tokens would make the attention backward a difference of nearly
        # equal terms, a badly conditioned target for a bf16 parity check
        feats = (2.0 * detweights.uniform(B * R * c["v_feature_size"], seed + 102) - 1.0).astype(np.float32).reshape(B, R, -1)
        bbox = detweights.uniform(B * R * 5, seed + 104).astype(np.float32).reshape(B, R, 5)
        max_features = np.array([R, R - 2, R - 1], dtype=np.int64)[:B]
        NB = B * B if c.get("in_batch_pairs", False) else B          # in_batch_pairs: one score row per (text, image) pair
        targets = np.zeros((NB, c["num_labels"]), dtype=np.float32)
        for b in range(NB):
            targets[b, (3 * b + 1) % c["num_labels"]] = 1.0
            targets[b, (5 * b + 2) % c["num_labels"]] = 0.6
        sl = SampleList(input_ids=torch.from_numpy(ids), input_mask=torch.from_numpy(mask), segment_ids=torch.from_numpy(seg),
                        image_feature_0=torch.from_numpy(feats),
                        image_info_0=SampleList(max_features=torch.from_numpy(max_features), bbox=torch.from_numpy(bbox)),
                        targets=torch.from_numpy(targets), dataset_name="vqa2", dataset_type="train")
        extra = {}
        if nlvr2:      # two images per sample (vilbert.py:1369-1394); labels are class indices
            feats1 = (2.0 * detweights.uniform(B * R * c["v_feature_size"], seed + 202) - 1.0).astype(np.float32).reshape(B, R, -1)
            bbox1 = detweights.uniform(B * R * 5, seed + 204).astype(np.float32).reshape(B, R, 5)
            mf1 = np.array([R - 1, R, R - 3], dtype=np.int64)[:B]
            labels = (detweights.uniform(B, seed + 203) > 0.5).astype(np.int64)
            sl = SampleList(input_ids=torch.from_numpy(ids), input_mask=torch.from_numpy(mask), segment_ids=torch.from_numpy(seg),
                            img0=SampleList(image_feature_0=torch.from_numpy(feats),
                                            image_info_0=SampleList(max_features=torch.from_numpy(max_features), bbox=torch.from_numpy(bbox))),
                            img1=SampleList(image_feature_0=torch.from_numpy(feats1),
                                            image_info_0=SampleList(max_features=torch.from_numpy(mf1), bbox=torch.from_numpy(bbox1))),
                            targets=torch.from_numpy(labels), dataset_name="nlvr2", dataset_type="train")
            extra = {"in_feats1": feats1, "in_bbox1": bbox1, "in_max_features1": mf1, "in_labels": labels}
        holder = {}
        orig = ref.model.bert.forward

        def spy(*a, **k):
            out = orig(*a, **k)
            holder["t"], holder["v"], holder["pt"], holder["pv"] = out[0], out[1], out[2], out[3]
            return out

        ref.model.bert.forward = spy
        if c.get("fast_mode", False):
            ids, mask, seg = ids[1:2], mask[1:2], seg[1:2]              # the half-padded text of sample 1, alone
            image_mask = (torch.arange(R).expand(B, R) < torch.from_numpy(max_features).unsqueeze(-1)).long()      # vilbert.py:1405-1413
            out = ref.model(torch.from_numpy(ids), torch.from_numpy(feats), torch.from_numpy(bbox), torch.from_numpy(seg), torch.from_numpy(mask), image_mask)
        else:
            out = ref(sl)
        if nlvr2:
            from mmf.modules.losses import CrossEntropyLoss
            loss = CrossEntropyLoss()(sl, out)
        else:
            loss = LogitBinaryCrossEntropy()(sl, out)
        loss.backward()
        rec = {"in_input_ids": ids, "in_input_mask": mask, "in_segment_ids": seg, "in_image_feature_0": feats, "in_bbox": bbox,
               "in_max_features": max_features, "in_targets": targets}
        rec.update(extra)
        rec["scores"] = out["scores"].detach().numpy()
        rec["sequence_output_t"] = holder["t"].detach().numpy()
        rec["sequence_output_v"] = holder["v"].detach().numpy()
        rec["pooled_output_t"] = holder["pt"].detach().numpy()
        rec["pooled_output_v"] = holder["pv"].detach().numpy()
        rec["loss"] = np.array(loss.item(), dtype=np.float64)
        names, norms, sums = [], [], []
        for k, p in ref.named_parameters():
            g = p.grad
            names.append(k)
            norms.append(0.0 if g is None else float(g.double().norm()))
            sums.append(0.0 if g is None else float(g.double().sum()))
            if g is not None and g.numel() <= 4096:
                rec["grad::" + k] = g.numpy()
        rec["grad_names"] = np.array(names)
        rec["grad_norms"] = np.array(norms)
        rec["grad_sums"] = np.array(sums)
        rec["param_names"] = np.array(list(shapes.keys()))
        rec["param_shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
        rec["case"] = np.array(repr(c))
        path = os.path.join(HERE, "%s.npz" % name)
        np.savez_compressed(path, **rec)
        print(name, "loss", loss.item(), "scores[0,:4]", rec["scores"][0, :4], "->", path, os.path.getsize(path), "bytes")


def make_vilbert_pretraining(visual_target=0):
    """`visual_target=1` (round 3) writes vilbert_pretraining_vt1.npz: the masked-region REGRESSION form (nn.MSELoss, vilbert.py:1074-1075,
    1139-1148) with regression targets in place of the class distributions; the first fixture is unchanged.
    ViLBERTForPretraining (vilbert.py:1054-1240, `visual_target: 0`) through the reference's own ViLBERT.forward /
    get_image_and_text_features, ViLBERTForPretraining.forward, vilbert.BertPreTrainingHeads (HF BertLMPredictionHead tied the
    pinned-transformers way + BertImagePredictionHead) and ViLBERTBase.  Only `ViLBERTBase.from_pretrained` (network) is
    replaced by constructing `ViLBERTBase(config)` directly."""
    from torch import nn
    from transformers import BertConfig
    M = refshim.ref_import("mmf.models.vilbert")
    M.replace_with_jit = lambda: None
    c = dict(VILBERT_CASES["vilbert_small"], seed=91, v_target_size=52 if visual_target == 2 else 53)
    torch.manual_seed(c["seed"])
    cfg = vilbert_reference_config(c)
    cfg["training_head_type"] = "pretraining"
    cfg["v_target_size"] = c["v_target_size"]
    cfg["losses"] = []
    bcfg = BertConfig.from_dict(OmegaConf.to_container(cfg))

    class RefPre(nn.Module):   # module tree of ViLBERTForPretraining (vilbert.py:1055-1077)
        def __init__(self):
            super().__init__()
            self.config = cfg
            self.bert = M.ViLBERTBase(bcfg)
            self.cls = M.BertPreTrainingHeads(bcfg)
            self.vocab_size = c["vocab_size"]
            self.visual_target = visual_target
            self.num_negative = 10 if visual_target == 2 else 128
            self.loss_fct = nn.CrossEntropyLoss(ignore_index=-1)
            self.vis_criterion = (nn.KLDivLoss(reduction="none") if visual_target == 0 else nn.MSELoss(reduction="none") if visual_target == 1
                                  else nn.CrossEntropyLoss())          # vilbert.py:1070-1075
            # tie_weights (:1088-1095) + transformers<=4.10 BertLMPredictionHead (decoder.bias IS predictions.bias)
            self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight
            self.cls.predictions.decoder.bias = self.cls.predictions.bias

        forward = M.ViLBERTForPretraining.forward

    class RefViLBERT(nn.Module):
        def __init__(self):
            super().__init__()
            self.config = cfg
            self.model = RefPre()

        get_image_and_text_features = M.ViLBERT.get_image_and_text_features
        forward = M.ViLBERT.forward

    ref = RefViLBERT().eval()
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()
              if not k.endswith("position_ids") and not k.endswith("embeddings.token_type_ids")
              and not k.startswith("model.cls.predictions.decoder.")}
    sd = detweights.state_dict(shapes, c["seed"])
    missing, unexpected = ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    B, T, R, seed = c["B"], c["T"], c["R"], c["seed"]
    ids = (detweights.uniform(B * T, seed + 100) * c["vocab_size"]).astype(np.int64).reshape(B, T)
    mask = np.ones((B, T), dtype=np.int64)
    mask[1, T // 2:] = 0
    mask[2, T - 3:] = 0
    ids[mask == 0] = 0
    seg = np.zeros((B, T), dtype=np.int64)
    feats = (2.0 * detweights.uniform(B * R * c["v_feature_size"], seed + 102) - 1.0).astype(np.float32).reshape(B, R, -1)
    bbox = detweights.uniform(B * R * 5, seed + 104).astype(np.float32).reshape(B, R, 5)
    max_features = np.array([R, R - 2, R - 1], dtype=np.int64)[:B]
    pick = (detweights.uniform(B * T, seed + 301).reshape(B, T) < 0.3) & (mask == 1)
    pick[:, 1] = True
    lm = np.where(pick, (detweights.uniform(B * T, seed + 302) * c["vocab_size"]).astype(np.int64).reshape(B, T), -1)
    # detector class distributions per region (sparse: exact zeros exercise the 0 * log 0 = 0 convention) and the masked regions
    raw = detweights.uniform(B * R * c["v_target_size"], seed + 303).reshape(B, R, -1)
    raw = np.where(raw < 0.6, 0.0, raw) ** 3
    raw[..., 0] += 1e-3
    cls_prob = (raw / raw.sum(-1, keepdims=True)).astype(np.float32)
    if visual_target == 1:       # regression targets (the reference reads them from the same `cls_prob` slot, vilbert.py:1402-1406)
        cls_prob = (2.0 * detweights.uniform(B * R * c["v_target_size"], seed + 305) - 1.0).astype(np.float32).reshape(B, R, -1)
    image_labels = (detweights.uniform(B * R, seed + 304).reshape(B, R) < 0.4).astype(np.int64)
    image_labels[:, 2] = 1
    image_labels[np.arange(R)[None, :] >= max_features[:, None]] = -1          # padded regions carry -1
    sl = SampleList(input_ids=torch.from_numpy(ids), input_mask=torch.from_numpy(mask), segment_ids=torch.from_numpy(seg),
                    image_feature_0=torch.from_numpy(feats),
                    image_info_0=SampleList(max_features=torch.from_numpy(max_features), bbox=torch.from_numpy(bbox), cls_prob=cls_prob),
                    image_labels=torch.from_numpy(image_labels), lm_label_ids=torch.from_numpy(lm), dataset_name="coco",
                    dataset_type="train")
    draws = []
    if visual_target == 2:
        # `visual_target: 2` samples its negatives with Tensor.random_ (vilbert.py:1158-1203): the draws are made deterministic and recorded,
        # so that the oracle and the HIP-backed model can be handed the same negatives
        cls_prob = (2.0 * detweights.uniform(B * R * c["v_target_size"], seed + 306) - 1.0).astype(np.float32).reshape(B, R, -1)
        sl["image_info_0"]["cls_prob"] = cls_prob
        real_random_ = torch.Tensor.random_

        def fake_random_(self, lo=0, hi=None):
            vals = np.minimum((detweights.uniform(self.numel(), seed + 900 + len(draws)) * (hi - lo)).astype(np.int64) + lo, hi - 1)
            draws.append(vals.reshape(tuple(self.shape)))
            return self.copy_(torch.from_numpy(draws[-1]))
        torch.Tensor.random_ = fake_random_
    try:
        out = ref(sl)
    finally:
        if visual_target == 2:
            torch.Tensor.random_ = real_random_
    losses = out["losses"]
    total = sum(v.sum() for v in losses.values())
    total.backward()
    rec = {"in_input_ids": ids, "in_input_mask": mask, "in_segment_ids": seg, "in_image_feature_0": feats, "in_bbox": bbox,
           "in_max_features": max_features, "in_lm_label_ids": lm, "in_cls_prob": cls_prob, "in_image_labels": image_labels}
    if visual_target == 2:
        assert len(draws) == 3, [d.shape for d in draws]
        for i, dr in enumerate(draws):
            rec["in_draw%d" % i] = dr
        # the flat negative indices those draws give (own restatement of :1158-1203; the oracle test checks it reproduces the recorded loss)
        ra, ca, ci = (d.copy() for d in draws)
        for i in range(B - 1):
            ra[i][ra[i] == i] = B - 1
        for i in range(R - 1):
            ci[:, i, :][ci[:, i, :] == i] = R - 1
        rec["in_negative_index"] = np.concatenate([ra * R + ca, np.arange(B).reshape(B, 1, 1) * R + ci], axis=2).astype(np.int64)
    rec["loss_keys"] = np.array(list(losses.keys()))
    rec["loss_values"] = np.array([float(v.sum()) for v in losses.values()], dtype=np.float64)
    rec["loss_shapes"] = np.array([",".join(map(str, v.shape)) for v in losses.values()])
    names, norms, sums = [], [], []
    for k, p in ref.named_parameters():
        g = p.grad
        names.append(k)
        norms.append(0.0 if g is None else float(g.double().norm()))
        sums.append(0.0 if g is None else float(g.double().sum()))
        if g is not None and g.numel() <= 4096:
            rec["grad::" + k] = g.numpy()
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array(norms)
    rec["grad_sums"] = np.array(sums)
    rec["param_names"] = np.array(list(shapes.keys()))
    rec["param_shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
    rec["state_dict_keys"] = np.array(sorted(k for k in ref.state_dict().keys()
                                             if not k.endswith("position_ids") and not k.endswith("embeddings.token_type_ids")))
    rec["case"] = np.array(repr(c))
    path = os.path.join(HERE, "vilbert_pretraining.npz" if visual_target == 0 else "vilbert_pretraining_vt%d.npz" % visual_target)
    np.savez_compressed(path, **rec)
    print("vilbert_pretraining", dict(zip(rec["loss_keys"], rec["loss_values"])), rec["loss_shapes"], "->", path, os.path.getsize(path), "bytes")


def make_transformer_heads():
    """The reference's `mlm` and `itm` transformer heads (mmf/models/transformers/heads/{mlm,itm}.py) run stand-alone on a fixed
    sequence output: logits, losses, parameter gradients and the gradient handed back to the encoder."""
    from torch import nn
    MLM = refshim.ref_import("mmf.models.transformers.heads.mlm").MLM
    ITM = refshim.ref_import("mmf.models.transformers.heads.itm").ITM
    c = dict(B=3, S=19, hidden_size=128, vocab_size=211, seed=95)
    B, S, H, V, seed = c["B"], c["S"], c["hidden_size"], c["vocab_size"], c["seed"]
    mlm = MLM(OmegaConf.create(dict(type="mlm", vocab_size=V, hidden_size=H))).eval()
    itm = ITM(OmegaConf.create(dict(type="itm", hidden_size=H))).eval()
    table = nn.Embedding(V, H)
    mlm.tie_weights(table)
    mlm.cls.predictions.decoder.bias = mlm.cls.predictions.bias        # transformers<=4.10 BertLMPredictionHead (the reference's pin)
    rec = {}
    for tag, mod in (("mlm", mlm), ("itm", itm), ("table", table)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items() if not k.startswith("cls.predictions.decoder.")}
        sd = detweights.state_dict({tag + "." + k: s for k, s in shapes.items()}, seed)
        missing, unexpected = mod.load_state_dict({k[len(tag) + 1:]: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not unexpected, unexpected
        rec[tag + "_param_names"] = np.array(list(sd.keys()))
        rec[tag + "_param_shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    assert mlm.cls.predictions.decoder.weight is table.weight
    seq = torch.from_numpy((2.0 * detweights.uniform(B * S * H, seed + 1) - 1.0).astype(np.float32).reshape(B, S, H)).requires_grad_(True)
    pick = detweights.uniform(B * S, seed + 2).reshape(B, S) < 0.25
    pick[:, 3] = True
    labels = np.where(pick, (detweights.uniform(B * S, seed + 3) * V).astype(np.int64).reshape(B, S), -1)
    is_correct = np.array([1, 0, 1], dtype=np.int64)[:B]
    proc = {"mlm_labels": {"combined_labels": torch.from_numpy(labels)}, "itm_labels": {"is_correct": torch.from_numpy(is_correct)}}
    out_mlm = mlm(seq, processed_sample_list=proc)
    out_itm = itm(seq, processed_sample_list=proc)
    total = out_mlm["losses"]["masked_lm_loss"] + out_itm["losses"]["itm_loss"]
    total.backward()
    rec.update({"in_sequence_output": seq.detach().numpy(), "in_labels": labels, "in_is_correct": is_correct,
                "mlm_logits": out_mlm["logits"].detach().numpy(), "mlm_loss": np.array(out_mlm["losses"]["masked_lm_loss"].item()),
                "itm_loss": np.array(out_itm["losses"]["itm_loss"].item()), "grad_sequence_output": seq.grad.numpy()})
    # MRC (masked region classification, heads/mrc.py): both loss variants on the same parameters
    MRC = refshim.ref_import("mmf.models.transformers.heads.mrc").MRC
    LD = 37
    mrc = MRC(hidden_size=H, label_dim=LD).eval()
    shapes = {k: tuple(v.shape) for k, v in mrc.state_dict().items()}
    sdm = detweights.state_dict({"mrc." + k: s_ for k, s_ in shapes.items()}, seed)
    mrc.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in sdm.items()}, strict=True)
    rec["mrc_param_names"] = np.array(list(sdm.keys()))
    rec["mrc_param_shapes"] = np.array([",".join(map(str, v.shape)) for v in sdm.values()])
    rmask = detweights.uniform(B * S, seed + 4).reshape(B, S) < 0.3
    rmask[:, 5] = True
    nm = int(rmask.sum())
    raw = detweights.uniform(nm * LD, seed + 5).reshape(nm, LD)
    raw = np.where(raw < 0.5, 0.0, raw) ** 2
    raw[:, 1] += 1e-3
    region_class = (raw / raw.sum(-1, keepdims=True)).astype(np.float32)
    proc2 = {"region_class": torch.from_numpy(region_class), "image_region_mask": torch.from_numpy(rmask)}
    seq2 = seq.detach().clone().requires_grad_(True)
    kl = mrc(seq2, proc2)["losses"]["mrc_loss"]
    kl.backward()
    rec.update({"in_region_mask": rmask, "in_region_class": region_class, "mrc_kl_loss": np.array(kl.item()),
                "mrc_kl_grad_sequence_output": seq2.grad.numpy().copy()})
    for k, p_ in mrc.named_parameters():
        rec["grad::mrc_kl." + k] = p_.grad.numpy().copy()
    mrc.zero_grad()
    mrc.use_kl = False
    seq3 = seq.detach().clone().requires_grad_(True)
    ce = mrc(seq3, proc2)["losses"]["mrc_loss"]
    ce.backward()
    rec.update({"mrc_ce_loss": np.array(ce.item()), "mrc_ce_grad_sequence_output": seq3.grad.numpy().copy()})
    for k, p_ in mrc.named_parameters():
        rec["grad::mrc_ce." + k] = p_.grad.numpy().copy()
    # MRFR (masked region feature regression) and WRA (word-region alignment by optimal transport): fixtures for the oracles that
    # precede their HIP implementation
    MRFR = refshim.ref_import("mmf.models.transformers.heads.mrfr").MRFR
    WRA = refshim.ref_import("mmf.models.transformers.heads.wra").WRA
    IMG = 40
    img_w = nn.Parameter(torch.from_numpy(detweights.state_dict({"img.weight": (H, IMG)}, seed)["img.weight"]))
    mrfr = MRFR(img_w, hidden_size=H, img_dim=IMG).eval()
    shapes = {k: tuple(v.shape) for k, v in mrfr.state_dict().items() if k != "linear_proj_weight"}
    sdf = detweights.state_dict({"mrfr." + k: s_ for k, s_ in shapes.items()}, seed)
    missing, unexpected = mrfr.load_state_dict({k[5:]: torch.from_numpy(v) for k, v in sdf.items()}, strict=False)
    assert not unexpected and missing == ["linear_proj_weight"], (missing, unexpected)
    rec["mrfr_param_names"] = np.array(list(sdf.keys()))
    rec["mrfr_param_shapes"] = np.array([",".join(map(str, v.shape)) for v in sdf.values()])
    rec["img_param_names"] = np.array(["img.weight"])
    rec["img_param_shapes"] = np.array(["%d,%d" % (H, IMG)])
    feat_targets = (2.0 * detweights.uniform(nm * IMG, seed + 6) - 1.0).astype(np.float32).reshape(nm, IMG)
    seq4 = seq.detach().clone().requires_grad_(True)
    lf = mrfr(seq4, {"mrfr_region_target": torch.from_numpy(feat_targets), "mrfr_region_mask": torch.from_numpy(rmask)})["losses"]["mrfr_loss"]
    lf.backward()
    rec.update({"in_mrfr_target": feat_targets, "mrfr_loss": np.array(lf.item()), "mrfr_grad_sequence_output": seq4.grad.numpy().copy(),
                "grad::img.weight": img_w.grad.numpy().copy()})
    for k, p_ in mrfr.named_parameters():
        if k != "linear_proj_weight":
            rec["grad::mrfr." + k] = p_.grad.numpy().copy()
    TL, IL = 12, 7                                   # S = 19 = 12 text rows + 7 region rows
    txt_pad = np.zeros((B, TL), dtype=bool); txt_pad[1, 8:] = True; txt_pad[2, 10:] = True
    img_pad = np.zeros((B, IL), dtype=bool); img_pad[0, 6:] = True; img_pad[2, 4:] = True
    wra = WRA().eval()
    seq5 = seq.detach().clone().requires_grad_(True)
    procw = {"wra_info": {"txt_pad": torch.from_numpy(txt_pad), "img_pad": torch.from_numpy(img_pad)},
             "is_correct": torch.from_numpy(is_correct), "input_ids": torch.zeros(B, TL, dtype=torch.long),
             "image_feat": torch.zeros(B, IL, 4)}
    lw = wra(seq5, procw)["losses"]["wra_loss"]
    lw.backward()
    rec.update({"in_txt_pad": txt_pad, "in_img_pad": img_pad, "wra_loss": np.array(lw.item()),
                "wra_grad_sequence_output": seq5.grad.numpy().copy()})
    for tag, mod in (("mlm", mlm), ("itm", itm)):
        seen = set()
        for k, p in mod.named_parameters():
            if id(p) in seen or p.grad is None:
                continue
            seen.add(id(p))
            rec["grad::%s.%s" % (tag, k)] = p.grad.numpy()
    rec["grad::table.weight"] = table.weight.grad.numpy()
    rec["mlm_state_dict_keys"] = np.array(sorted(mlm.state_dict().keys()))
    rec["itm_state_dict_keys"] = np.array(sorted(itm.state_dict().keys()))
    rec["case"] = np.array(repr(c))
    path = os.path.join(HERE, "transformer_heads.npz")
    np.savez_compressed(path, **rec)
    print("transformer_heads mrfr", float(rec["mrfr_loss"]), "wra", float(rec["wra_loss"]), "mrc kl / ce", float(rec["mrc_kl_loss"]), float(rec["mrc_ce_loss"]), "mlm_loss", float(rec["mlm_loss"]), "itm_loss", float(rec["itm_loss"]), "logits", rec["mlm_logits"].shape, "->", path,
          os.path.getsize(path), "bytes")


UNITER_CASES = {
    "uniter_small64": dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=211,
                           max_position_embeddings=40, img_dim=72, head_hidden_size=256, num_labels=13, B=3, T=12, R=7, seed=51),
}


def make_uniter():
    """UNITER (classification, task vqa2) through the reference's own UNITER.forward / add_custom_params / add_pos_feat,
    UNITERForClassification.forward -> _infer_with_heads, UNITERModelBase.forward + _compute_*_embeddings,
    UNITERImageEmbeddings, HF BertEmbeddings / BertEncoder and the MLP head (mmf/models/uniter.py).  Only the
    `from_pretrained` calls of UNITERModelBase.__init__ (network) are replaced by constructing the same HF classes from a
    small config."""
    from torch import nn
    from transformers import BertConfig
    from transformers.models.bert.modeling_bert import BertEmbeddings, BertModel
    MLP = refshim.ref_import("mmf.models.transformers.heads.mlp").MLP
    U = refshim.ref_import("mmf.models.uniter")
    from mmf.modules.losses import MMFLoss

    for name, c in UNITER_CASES.items():
        torch.manual_seed(c["seed"])
        bcfg = BertConfig(hidden_size=c["hidden_size"], num_hidden_layers=c["num_hidden_layers"],
                          num_attention_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
                          vocab_size=c["vocab_size"], max_position_embeddings=c["max_position_embeddings"], type_vocab_size=2,
                          hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12, pad_token_id=0)
        bcfg._attn_implementation = "eager"

        class EncoderCompat(nn.Module):
            """HF BertEncoder called the transformers<=4.10 way (the reference's pin): positional tuple output
            (last_hidden_state, all_hidden_states).  The installed transformers 5 no longer returns the hidden states from
            the encoder call, so the layer loop of BertEncoder.forward is spelled out over the very same BertLayer modules."""

            def __init__(self, enc):
                super().__init__()
                self.layer = enc.layer

            def forward(self, hidden_states, attention_mask=None, output_hidden_states=False):
                all_hidden = ()
                for layer in self.layer:
                    all_hidden = all_hidden + (hidden_states,)
                    out = layer(hidden_states, attention_mask=attention_mask)
                    hidden_states = out[0] if isinstance(out, (tuple, list)) else out
                all_hidden = all_hidden + (hidden_states,)
                return (hidden_states, all_hidden)

        class RefBase(nn.Module):   # module tree of UNITERModelBase (uniter.py:117-150)
            def __init__(self):
                super().__init__()
                self.text_embeddings = BertEmbeddings(bcfg)
                self.img_embeddings = U.UNITERImageEmbeddings(img_dim=c["img_dim"], hidden_size=c["hidden_size"], hidden_dropout_prob=0.1)
                bm = BertModel(bcfg)
                self.encoder = EncoderCompat(bm.encoder)
                self.pooler = bm.pooler

        for fn in ("_compute_txt_embeddings", "_compute_img_embeddings", "_compute_img_txt_embeddings", "forward"):
            setattr(RefBase, fn, getattr(U.UNITERModelBase, fn))

        head_cfg = OmegaConf.create(dict(type="mlp", freeze=False, lr_multiplier=1.0, in_dim=c["hidden_size"],
                                         hidden_size=c["head_hidden_size"], num_labels=c["num_labels"], pooler_name="bert_pooler"))

        class RefCls(nn.Module):    # UNITERForClassification (uniter.py:286-347)
            def __init__(self):
                super().__init__()
                self.uniter = RefBase()
                self.heads = nn.ModuleDict({"vqa2": MLP(head_cfg)})
                self.tasks = ["vqa2"]
                self.losses = nn.ModuleDict({"vqa2": MMFLoss("logit_bce")})

            forward = U.UNITERForClassification.forward

        class RefUNITER(nn.Module):  # the registered BaseModel (uniter.py:621-773)
            def __init__(self):
                super().__init__()
                self.uniter = RefCls()
                self.tasks = ["vqa2"]

            add_pos_feat = U.UNITER.add_pos_feat
            add_custom_params = U.UNITER.add_custom_params
            forward = U.UNITER.forward

        ref = RefUNITER().eval()
        shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()
                  if not k.endswith("position_ids") and not k.endswith("embeddings.token_type_ids")}
        sd = detweights.state_dict(shapes, c["seed"])
        missing, unexpected = ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not unexpected, unexpected
        B, T, R, seed = c["B"], c["T"], c["R"], c["seed"]
        ids = (detweights.uniform(B * T, seed + 100) * c["vocab_size"]).astype(np.int64).reshape(B, T)
        mask = np.ones((B, T), dtype=np.int64)
        mask[1, T // 2:] = 0
        mask[2, T - 3:] = 0
        ids[mask == 0] = 0   # [PAD]
        feats = (2.0 * detweights.uniform(B * R * c["img_dim"], seed + 102) - 1.0).astype(np.float32).reshape(B, R, -1)
        xy = detweights.uniform(B * R * 4, seed + 104).astype(np.float32).reshape(B, R, 4)
        x1 = np.minimum(xy[..., 0], xy[..., 2]) * 600 + 2; x2 = np.maximum(xy[..., 0], xy[..., 2]) * 600 + 4
        y1 = np.minimum(xy[..., 1], xy[..., 3]) * 400 + 2; y2 = np.maximum(xy[..., 1], xy[..., 3]) * 400 + 4
        bbox = np.stack([x1, y1, x2, y2], axis=-1).astype(np.float32)      # pixel boxes -> normalised by (w, h) in add_pos_feat
        img_w = np.full((B,), 640, dtype=np.int64); img_h = np.full((B,), 480, dtype=np.int64)
        max_features = np.array([R, R - 2, R - 1], dtype=np.int64)[:B]
        targets = np.zeros((B, c["num_labels"]), dtype=np.float32)
        for b in range(B):
            targets[b, (3 * b + 1) % c["num_labels"]] = 1.0
            targets[b, (5 * b + 2) % c["num_labels"]] = 0.6
        sl = SampleList(input_ids=torch.from_numpy(ids), input_mask=torch.from_numpy(mask), segment_ids=torch.zeros(B, T, dtype=torch.long),
                        image_feature_0=torch.from_numpy(feats),
                        image_info_0=SampleList(max_features=torch.from_numpy(max_features), bbox=bbox, image_width=img_w, image_height=img_h),
                        targets=torch.from_numpy(targets), dataset_name="vqa2", dataset_type="train")
        holder = {}
        orig = ref.uniter.uniter.forward

        def spy(*a, **k):
            out = orig(*a, **k)
            holder["seq"] = out.final_layer
            return out

        ref.uniter.uniter.forward = spy
        out = ref(sl)
        (lkey, loss), = out["losses"].items()
        loss = loss.sum()
        loss.backward()
        rec = {"in_input_ids": ids, "in_input_mask": mask, "in_image_feature_0": feats, "in_bbox": bbox, "in_image_width": img_w,
               "in_image_height": img_h, "in_max_features": max_features, "in_targets": targets}
        rec["scores"] = out["scores"].detach().numpy()
        rec["sequence_output"] = holder["seq"].detach().numpy()
        rec["img_pos_feat"] = sl["img_pos_feat"].detach().numpy()
        rec["loss"] = np.array(loss.item(), dtype=np.float64)
        rec["loss_key"] = np.array(lkey)
        names, norms, sums = [], [], []
        for k, p in ref.named_parameters():
            g = p.grad
            names.append(k)
            norms.append(0.0 if g is None else float(g.double().norm()))
            sums.append(0.0 if g is None else float(g.double().sum()))
            if g is not None and g.numel() <= 4096:
                rec["grad::" + k] = g.numpy()
        rec["grad_names"] = np.array(names)
        rec["grad_norms"] = np.array(norms)
        rec["grad_sums"] = np.array(sums)
        rec["param_names"] = np.array(list(shapes.keys()))
        rec["param_shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
        rec["case"] = np.array(repr(c))
        path = os.path.join(HERE, "%s.npz" % name)
        np.savez_compressed(path, **rec)
        print(name, "loss", loss.item(), lkey, "scores[0,:4]", rec["scores"][0, :4], "->", path, os.path.getsize(path), "bytes")


M4C_CASES = {
    # text_bert 128-wide (2 heads) projected to a 192-wide (3 heads) MMT; ragged text / objects / OCR; 5 decoding steps
    "m4c_small64": dict(text_hidden_size=128, text_num_hidden_layers=2, text_num_attention_heads=2, text_intermediate_size=256,
                        vocab_size=211, max_position_embeddings=40, hidden_size=192, num_hidden_layers=2, num_attention_heads=3,
                        intermediate_size=384, obj_in_dim=72, obj_fc7_dim=80, ocr_in_dim=72, ocr_fc7_dim=88, num_choices=37,
                        query_key_size=128, B=3, T=8, O=7, N=6, D=5, seed=71),
}


def m4c_inputs(c):
    """Deterministic M4C batch (numpy): the keys M4C.forward reads from the sample list (m4c.py:183-253,285-294) plus
    the loss inputs (losses.py:581-592)."""
    B, T, O, N, D, seed = c["B"], c["T"], c["O"], c["N"], c["D"], c["seed"]
    V = c["num_choices"]
    u = detweights.uniform
    text = (u(B * T, seed + 100) * c["vocab_size"]).astype(np.int64).reshape(B, T)
    text_len = np.array([T, T // 2, T - 2], dtype=np.int64)[:B]
    for b in range(B):
        text[b, text_len[b]:] = 0     # [PAD]
    rec = dict(
        text=text, text_len=text_len,
        image_feature_0=(2.0 * u(B * O * c["obj_in_dim"], seed + 101) - 1.0).astype(np.float32).reshape(B, O, -1),
        obj_bbox_coordinates=u(B * O * 4, seed + 102).astype(np.float32).reshape(B, O, 4),
        obj_max_features=np.array([O, O - 2, O - 1], dtype=np.int64)[:B],
        context_feature_0=(2.0 * u(B * N * 300, seed + 103) - 1.0).astype(np.float32).reshape(B, N, 300),
        context_feature_1=u(B * N * 604, seed + 104).astype(np.float32).reshape(B, N, 604),
        image_feature_1=(2.0 * u(B * (N + 3) * c["ocr_in_dim"], seed + 105) - 1.0).astype(np.float32).reshape(B, N + 3, -1),
        ocr_bbox_coordinates=u(B * N * 4, seed + 106).astype(np.float32).reshape(B, N, 4),
        ocr_max_features=np.array([N, N - 3, N - 2], dtype=np.int64)[:B],
        order_vectors=u(B * N * N, seed + 107).astype(np.float32).reshape(B, N, N),   # zeroed by the model (m4c.py:227)
    )
    prev = (u(B * D, seed + 108) * (V + N)).astype(np.int64).reshape(B, D)
    prev[:, 0] = 1            # BOS
    prev[1, 3:] = 0           # <pad> repeated: colliding rows in the gather's backward
    prev[2, 2] = V + 1        # an OCR copy
    prev[0, 4] = V + N - 1
    rec["train_prev_inds"] = prev
    tg = np.zeros((B, D, V + N), dtype=np.float32)
    pick = (u(B * D * 2, seed + 109) * (V + N)).astype(np.int64).reshape(B, D, 2)
    for b in range(B):
        for t in range(D):
            tg[b, t, pick[b, t, 0]] = 1.0
            tg[b, t, pick[b, t, 1]] = 0.6
    rec["targets"] = tg
    lm = np.ones((B, D), dtype=np.float32)
    lm[1, 3:] = 0.0
    lm[2, 4:] = 0.0
    rec["train_loss_mask"] = lm
    return rec


def m4c_sample_list(rec):
    t = torch.from_numpy
    return SampleList(
        text=t(rec["text"]), text_len=t(rec["text_len"]), image_feature_0=t(rec["image_feature_0"]),
        obj_bbox_coordinates=t(rec["obj_bbox_coordinates"]), image_info_0=SampleList(max_features=t(rec["obj_max_features"])),
        context_feature_0=t(rec["context_feature_0"]), context_feature_1=t(rec["context_feature_1"]),
        image_feature_1=t(rec["image_feature_1"]), ocr_bbox_coordinates=t(rec["ocr_bbox_coordinates"]),
        context_info_0=SampleList(max_features=t(rec["ocr_max_features"])), order_vectors=t(rec["order_vectors"]),
        train_prev_inds=t(rec["train_prev_inds"]), targets=t(rec["targets"]), train_loss_mask=t(rec["train_loss_mask"]),
        dataset_name="textvqa", dataset_type="train")


def make_m4c():
    """M4C (BASELINE configs[4]) through the reference's own `M4C._forward_*` methods over its `TextBert`, `MMT`,
    `OcrPtrNet`, `PrevPredEmbeddings`, `ClassifierLayer("linear")` and the `FinetuneFasterRcnnFpnFc7.forward` body, plus
    `M4CDecodingBCEWithMaskLoss`.  The registered class itself needs the dataset registry and the detectron fc7 pickles
    (m4c.py:36-44,56-68), so the module tree of M4C.build() is assembled here and the reference's methods are bound to it."""
    from torch import nn
    from transformers import BertConfig
    M = refshim.ref_import("mmf.models.m4c")
    E = refshim.ref_import("mmf.modules.encoders")
    from mmf.modules.layers import ClassifierLayer
    from mmf.modules.losses import M4CDecodingBCEWithMaskLoss

    def build_ref(c):
        tcfg = BertConfig(hidden_size=c["text_hidden_size"], num_hidden_layers=c["text_num_hidden_layers"],
                          num_attention_heads=c["text_num_attention_heads"], intermediate_size=c["text_intermediate_size"],
                          vocab_size=c["vocab_size"], max_position_embeddings=c["max_position_embeddings"], type_vocab_size=2,
                          hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12, pad_token_id=0)
        mcfg = BertConfig(hidden_size=c["hidden_size"], num_hidden_layers=c["num_hidden_layers"],
                          num_attention_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
                          hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12)
        tcfg._attn_implementation = "eager"
        mcfg._attn_implementation = "eager"
        H, N = c["hidden_size"], c["N"]

        class Fc7(nn.Module):       # FinetuneFasterRcnnFpnFc7 minus the pickle loading of its constructor
            def __init__(self, in_dim, out_dim):
                super().__init__()
                self.lc = nn.Linear(in_dim, out_dim)

            forward = E.FinetuneFasterRcnnFpnFc7.forward

        class RefM4C(nn.Module):    # module tree of M4C.build(), m4c.py:46-170
            def __init__(self):
                super().__init__()
                self.mmt_config = mcfg
                self.text_bert = M.TextBert(tcfg)
                self.text_bert_out_linear = nn.Linear(c["text_hidden_size"], H)
                self.obj_faster_rcnn_fc7 = Fc7(c["obj_in_dim"], c["obj_fc7_dim"])
                self.linear_obj_feat_to_mmt_in = nn.Linear(c["obj_fc7_dim"], H)
                self.linear_obj_bbox_to_mmt_in = nn.Linear(4, H)
                self.obj_feat_layer_norm = nn.LayerNorm(H)
                self.obj_bbox_layer_norm = nn.LayerNorm(H)
                self.obj_drop = nn.Dropout(0.1)
                self.remove_ocr_fasttext = self.remove_ocr_phoc = self.remove_ocr_frcn = False
                self.remove_ocr_semantics = self.remove_ocr_bbox = False
                self.ocr_faster_rcnn_fc7 = Fc7(c["ocr_in_dim"], c["ocr_fc7_dim"])
                self.linear_ocr_feat_to_mmt_in = nn.Linear(300 + 604 + c["ocr_fc7_dim"] + N, H)
                self.linear_ocr_bbox_to_mmt_in = nn.Linear(4, H)
                self.ocr_feat_layer_norm = nn.LayerNorm(H)
                self.ocr_bbox_layer_norm = nn.LayerNorm(H)
                self.ocr_drop = nn.Dropout(0.1)
                self.mmt = M.MMT(mcfg)
                self.ocr_ptr_net = M.OcrPtrNet(hidden_size=H, query_key_size=c["query_key_size"])
                self.classifier = ClassifierLayer("linear", in_dim=H, out_dim=c["num_choices"])
                self.answer_processor = SampleList(BOS_IDX=1)

        for fn in ("forward", "_forward_txt_encoding", "_forward_obj_encoding", "_forward_ocr_encoding", "_forward_mmt",
                   "_forward_output", "_forward_mmt_and_output"):
            setattr(RefM4C, fn, getattr(M.M4C, fn))

        ref = RefM4C().eval()
        shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()
                  if not k.endswith("position_ids") and not k.endswith("embeddings.token_type_ids")}
        sd = detweights.state_dict(shapes, c["seed"])
        for k in sd:               # nn.LayerNorm gains that are not called "LayerNorm.weight"
            if k.endswith("layer_norm.weight"):
                sd[k] = sd[k] + 1.0
        missing, unexpected = ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not unexpected, unexpected

        return ref, sd, shapes

    for name, c in M4C_CASES.items():
        ref, sd, shapes = build_ref(c)
        H, N = c["hidden_size"], c["N"]
        rec_in = m4c_inputs(c)
        sl = m4c_sample_list(rec_in)

        # teacher forcing (self.training, m4c.py:286-289) with every dropout off: only the top-level flag is set, the
        # sub-modules stay in eval mode
        ref.training = True
        fwd = {}
        ref._forward_txt_encoding(sl, fwd)
        ref._forward_obj_encoding(sl, fwd)
        ref._forward_ocr_encoding(sl, fwd)
        ref._forward_mmt_and_output(sl, fwd)
        out = {"scores": fwd["scores"]}
        loss = M4CDecodingBCEWithMaskLoss()(sl, out).sum()
        loss.backward()
        rec = {"in_" + k: v for k, v in rec_in.items()}
        rec["scores"] = out["scores"].detach().numpy()
        for k in ("obj_mmt_in", "ocr_mmt_in", "txt_emb", "mmt_seq_output"):
            rec[k] = fwd[k].detach().numpy()
        rec["loss"] = np.array(loss.item(), dtype=np.float64)
        names, norms, sums = [], [], []
        for k, p in ref.named_parameters():
            g = p.grad
            names.append(k)
            norms.append(0.0 if g is None else float(g.double().norm()))
            sums.append(0.0 if g is None else float(g.double().sum()))
            if g is not None and g.numel() <= 4096:
                rec["grad::" + k] = g.numpy()
        rec["grad_names"] = np.array(names)
        rec["grad_norms"] = np.array(norms)
        rec["grad_sums"] = np.array(sums)

        # greedy decoding (not self.training, m4c.py:290-305) through M4C.forward itself
        ref.training = False
        with torch.no_grad():
            dec = ref.forward(m4c_sample_list(rec_in))
        rec["decode_scores"] = dec["scores"].numpy()
        rec["decode_argmax"] = dec["scores"].argmax(dim=-1).numpy()
        top2 = dec["scores"].topk(2, dim=-1).values
        rec["decode_margin"] = (top2[..., 0] - top2[..., 1]).numpy()

        # second decoding fixture: with the deterministic weights above every step decodes BOS, so the loop's feedback path
        # (previous predictions >= num_choices select OCR rows in the two-source gather, m4c.py:526-528) is never taken.
        # detweights.sharpen_m4c_decoder rescales the output layers (no new randomness) so that the greedy sequence mixes
        # fixed-vocabulary and OCR-copy indices with margins far above bf16 noise.
        sd2 = detweights.sharpen_m4c_decoder(sd)
        ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()}, strict=False)
        with torch.no_grad():
            dec2 = ref.forward(m4c_sample_list(rec_in))
        rec["decode2_scores"] = dec2["scores"].numpy()
        rec["decode2_argmax"] = dec2["scores"].argmax(dim=-1).numpy()
        top2 = dec2["scores"].topk(2, dim=-1).values
        rec["decode2_margin"] = (top2[..., 0] - top2[..., 1]).numpy()

        rec["param_names"] = np.array(list(shapes.keys()))
        rec["param_shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
        rec["case"] = np.array(repr(c))
        path = os.path.join(HERE, "%s.npz" % name)
        np.savez_compressed(path, **rec)
        print(name, "loss", loss.item(), "scores[0,0,:4]", rec["scores"][0, 0, :4], "decode", rec["decode_argmax"].tolist(), "decode2", rec["decode2_argmax"].tolist(), "min margin", float(rec["decode2_margin"].min()),
              "->", path, os.path.getsize(path), "bytes")


def make_visual_bert_nlvr2():
    """VisualBERT with `training_head_type: nlvr2` (two images per sample, default BertPooler strategy) through the
    reference's own VisualBERT.forward (visual_bert.py:483-601) and VisualBERTForClassification (:284-404)."""
    from mmf.modules.losses import CrossEntropyLoss
    c = dict(CASES["small64"], num_labels=2, seed=61)
    cfg = reference_config(c)
    cfg["training_head_type"] = "nlvr2"
    cfg["pooler_strategy"] = "default"
    cfg["losses"] = [dict(type="cross_entropy")]
    model = ref_vb.VisualBERT(cfg)
    model.build()
    model.eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("position_ids")}
    sd = detweights.state_dict(shapes, c["seed"])
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("position_ids") for k in missing), (missing, unexpected)
    inp = make_inputs(c)
    B, R = c["B"], c["R"]
    feats1 = (detweights.uniform(B * R * c["visual_embedding_dim"], c["seed"] + 202)).astype(np.float32).reshape(B, R, -1)
    mf1 = np.array([R - 1, R, R - 3], dtype=np.int64)[:B]
    targets = (detweights.uniform(B, c["seed"] + 203) > 0.5).astype(np.int64)
    sl = SampleList(
        input_ids=torch.from_numpy(inp["input_ids"]), input_mask=torch.from_numpy(inp["input_mask"]),
        segment_ids=torch.from_numpy(inp["segment_ids"]),
        img0=SampleList(image_feature_0=torch.from_numpy(inp["image_feature_0"]),
                        image_info_0=SampleList(max_features=torch.from_numpy(inp["max_features"]))),
        img1=SampleList(image_feature_0=torch.from_numpy(feats1), image_info_0=SampleList(max_features=torch.from_numpy(mf1))),
        image_feature_0=torch.from_numpy(inp["image_feature_0"]),
        targets=torch.from_numpy(targets), dataset_name="nlvr2", dataset_type="train")
    out = model.forward(sl)
    loss = CrossEntropyLoss()(sl, out)
    loss.backward()
    rec = {"in_input_ids": inp["input_ids"], "in_input_mask": inp["input_mask"], "in_segment_ids": inp["segment_ids"],
           "in_feats0": inp["image_feature_0"], "in_feats1": feats1, "in_max_features0": inp["max_features"], "in_max_features1": mf1,
           "in_targets": targets}
    rec["scores"] = out["scores"].detach().numpy()
    rec["loss"] = np.array(loss.item(), dtype=np.float64)
    names, norms, sums = [], [], []
    for k, p in model.named_parameters():
        g = p.grad
        names.append(k)
        norms.append(0.0 if g is None else float(g.double().norm()))
        sums.append(0.0 if g is None else float(g.double().sum()))
        if g is not None and g.numel() <= 4096:
            rec["grad::" + k] = g.numpy()
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array(norms)
    rec["grad_sums"] = np.array(sums)
    rec["param_names"] = np.array(list(shapes.keys()))
    rec["param_shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
    rec["case"] = np.array(repr(c))
    path = os.path.join(HERE, "visual_bert_nlvr2.npz")
    np.savez_compressed(path, **rec)
    print("visual_bert_nlvr2 loss", loss.item(), "scores", rec["scores"], "->", path, os.path.getsize(path), "bytes")


def make_visual_bert_pretraining():
    """VisualBERT with `training_head_type: pretraining` (masked-LM head over the joint sequence, decoder tied to the word
    embeddings) through the reference's own VisualBERT.forward (visual_bert.py:567-601) and VisualBERTForPretraining (:160-281).
    Two records: mixed labels (loss, logits, every gradient) and all labels -1 (the reference's own test asserts NaN,
    tests/models/test_visual_bert.py:71-98)."""
    c = dict(CASES["small64"], seed=71)
    cfg = reference_config(c)
    cfg["training_head_type"] = "pretraining"
    cfg["losses"] = []
    model = ref_vb.VisualBERT(cfg)
    model.build()
    model.eval()
    heads = model.model.cls.predictions
    # transformers<=4.10 BertLMPredictionHead.__init__ (the reference's pin): `self.decoder.bias = self.bias` — one tensor under
    # two state-dict keys; 5.15 keeps two separate parameters, so tie them the pinned way before loading weights
    heads.decoder.bias = heads.bias
    assert heads.decoder.weight is model.model.bert.embeddings.word_embeddings.weight      # tie_weights, visual_bert.py:227-235
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()
              if not k.endswith("position_ids") and not k.startswith("model.cls.predictions.decoder.")}
    sd = detweights.state_dict(shapes, c["seed"])
    full = {k: torch.from_numpy(v) for k, v in sd.items()}
    full["model.cls.predictions.decoder.weight"] = full["model.bert.embeddings.word_embeddings.weight"]
    full["model.cls.predictions.decoder.bias"] = full["model.cls.predictions.bias"]
    missing, unexpected = model.load_state_dict(full, strict=False)
    assert not unexpected and all(k.endswith("position_ids") for k in missing), (missing, unexpected)
    inp = make_inputs(c)
    B, T = c["B"], c["T"]
    pick = detweights.uniform(B * T, c["seed"] + 301).reshape(B, T) < 0.3
    pick &= inp["input_mask"] == 1
    pick[:, 1] = True                                   # every sample has at least one masked position
    lm = np.where(pick, (detweights.uniform(B * T, c["seed"] + 302) * c["vocab_size"]).astype(np.int64).reshape(B, T), -1)

    def sample(labels):
        return SampleList(
            input_ids=torch.from_numpy(inp["input_ids"]), input_mask=torch.from_numpy(inp["input_mask"]),
            segment_ids=torch.from_numpy(inp["segment_ids"]), image_feature_0=torch.from_numpy(inp["image_feature_0"]),
            image_info_0=SampleList(max_features=torch.from_numpy(inp["max_features"])),
            lm_label_ids=torch.from_numpy(labels), dataset_name="coco", dataset_type="train")

    out = model.forward(sample(lm))
    (key, loss), = out["losses"].items()
    loss.backward()
    rec = {"in_" + k: v for k, v in inp.items() if k != "targets"}
    rec["in_lm_label_ids"] = lm
    rec["logits"] = out["logits"].detach().numpy()
    rec["sequence_output"] = out["sequence_output"].detach().numpy()
    rec["loss"] = np.array(loss.item(), dtype=np.float64)
    rec["loss_key"] = np.array(key)
    names, norms, sums = [], [], []
    for k, p in model.named_parameters():
        g = p.grad
        names.append(k)
        norms.append(0.0 if g is None else float(g.double().norm()))
        sums.append(0.0 if g is None else float(g.double().sum()))
        if g is not None and g.numel() <= 4096:
            rec["grad::" + k] = g.numpy()
    rec["grad::model.bert.embeddings.word_embeddings.weight"] = model.model.bert.embeddings.word_embeddings.weight.grad.numpy()
    with torch.no_grad():
        out2 = model.forward(sample(np.full((B, T), -1, dtype=np.int64)))
    rec["loss_all_ignored_is_nan"] = np.array(bool(torch.isnan(out2["losses"][key])))
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array(norms)
    rec["grad_sums"] = np.array(sums)
    rec["param_names"] = np.array(list(shapes.keys()))
    rec["param_shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
    rec["state_dict_keys"] = np.array(sorted(k for k in model.state_dict().keys() if not k.endswith("position_ids")))
    rec["case"] = np.array(repr(c))
    path = os.path.join(HERE, "visual_bert_pretraining.npz")
    np.savez_compressed(path, **rec)
    print("visual_bert_pretraining loss", loss.item(), key, "all-ignored NaN:", bool(rec["loss_all_ignored_is_nan"]), "->", path,
          os.path.getsize(path), "bytes")


if __name__ == "__main__":
    # no arguments: EVERY fixture this script owns (vilbert = all of VILBERT_CASES incl. dyn / fixed / pairs / fast + nlvr2)
    which = sys.argv[1:] or ["visual_bert", "alignment", "nlvr2", "pretraining", "mmbt", "mmbt_pretraining", "mmft", "vilbert", "vilbert_pretraining",
                             "vilbert_pretraining_vt2", "heads", "uniter", "m4c"]
    if "visual_bert" in which:
        main()
    if "alignment" in which:
        main(align=True)
    if "nlvr2" in which:
        make_visual_bert_nlvr2()
    if "pretraining" in which:
        make_visual_bert_pretraining()
    if "mmbt" in which:
        make_mmbt()
    if "mmbt_pretraining" in which:
        make_mmbt_pretraining()
    if "mmft" in which:
        make_mmft()
    if "vilbert" in which:
        make_vilbert()
    if "vilbert_fixed" in which:
        make_vilbert(only=("vilbert_fixed",))
    if "vilbert_pairs" in which:
        make_vilbert(only=("vilbert_pairs",))
    if "vilbert_fast" in which:
        make_vilbert(only=("vilbert_fast",))
    if "vilbert_pretraining_vt2" in which:
        make_vilbert_pretraining(visual_target=2)
    if "vilbert_pretraining" in which:
        make_vilbert_pretraining()
        make_vilbert_pretraining(visual_target=1)
    if "heads" in which:
        make_transformer_heads()
    if "uniter" in which:
        make_uniter()
    if "m4c" in which:
        make_m4c()
