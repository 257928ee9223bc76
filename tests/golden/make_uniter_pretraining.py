"""Generate tests/golden/uniter_pretraining.npz by running the ACTUAL reference `UNITERForPretraining.forward` (mmf/models/uniter.py:
350-618) for the tasks mlm, itm and mrc: its own preprocessing (`_process_sample_list_for_pretraining`, `_add_image_feat_masked`,
`_preprocess_{mlm,itm,mrc}`, `_mask_inputs_in_sample_list`, `_remove_mismatched_captions`), `_infer_with_heads`, `UNITERModelBase`
and the reference `mlm` / `itm` / `mrc` heads.  As in make_golden.py::make_uniter only the `from_pretrained` downloads of
UNITERModelBase.__init__ are replaced (the same HF classes built from a small config); the random region masks the reference draws
(`_get_img_mask`: numpy binomial + random.choice) are drawn under fixed seeds and RECORDED, so that the oracle and the HIP path can be
handed the same masks.

`--all-tasks` (round 3) writes a SECOND fixture, tests/golden/uniter_pretraining_all.npz, from the wrapper built with the reference's DEFAULT
task list mlm, itm, mrc, mrfr, wra (uniter.py:36-39; the `mrfr` head tied to `img_embeddings.img_linear.weight`, :397-400) and records
the two tasks the first fixture lacks: `_preprocess_mrfr` / `_preprocess_wra`, the MRFR and WRA heads (mmf/models/transformers/heads/
{mrfr,wra}.py over mmf/modules/ot.py).  The first fixture stays as it is (its parameter list is part of what it pins).

    python tests/golden/make_uniter_pretraining.py [--all-tasks]
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG  # noqa: E402  (installs the reference import shims)

refshim, detweights, OmegaConf, SampleList = MG.refshim, MG.detweights, MG.OmegaConf, MG.SampleList

CASE = dict(MG.UNITER_CASES["uniter_small64"], seed=97, label_dim=29, mask_probability=0.3)


def main(all_tasks=False):
    from torch import nn
    from transformers import BertConfig
    from transformers.models.bert.modeling_bert import BertEmbeddings, BertModel
    U = refshim.ref_import("mmf.models.uniter")
    MLM = refshim.ref_import("mmf.models.transformers.heads.mlm").MLM
    ITM = refshim.ref_import("mmf.models.transformers.heads.itm").ITM
    MRC = refshim.ref_import("mmf.models.transformers.heads.mrc").MRC
    MRFR = refshim.ref_import("mmf.models.transformers.heads.mrfr").MRFR
    WRA = refshim.ref_import("mmf.models.transformers.heads.wra").WRA
    c = CASE
    torch.manual_seed(c["seed"])
    H, V = c["hidden_size"], c["vocab_size"]
    bcfg = BertConfig(hidden_size=H, num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                      intermediate_size=c["intermediate_size"], vocab_size=V, max_position_embeddings=c["max_position_embeddings"],
                      type_vocab_size=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12, pad_token_id=0)
    bcfg._attn_implementation = "eager"

    class EncoderCompat(nn.Module):          # see make_golden.py::make_uniter
        def __init__(self, enc):
            super().__init__()
            self.layer = enc.layer

        def forward(self, hidden_states, attention_mask=None, output_hidden_states=False):
            all_hidden = ()
            for layer in self.layer:
                all_hidden = all_hidden + (hidden_states,)
                out = layer(hidden_states, attention_mask=attention_mask)
                hidden_states = out[0] if isinstance(out, (tuple, list)) else out
            return (hidden_states, all_hidden + (hidden_states,))

    class RefBase(nn.Module):                # module tree of UNITERModelBase (uniter.py:117-150)
        def __init__(self):
            super().__init__()
            self.text_embeddings = BertEmbeddings(bcfg)
            self.img_embeddings = U.UNITERImageEmbeddings(img_dim=c["img_dim"], hidden_size=H, hidden_dropout_prob=0.1)
            bm = BertModel(bcfg)
            self.encoder = EncoderCompat(bm.encoder)
            self.pooler = bm.pooler

    for fn in ("_compute_txt_embeddings", "_compute_img_embeddings", "_compute_img_txt_embeddings", "forward"):
        setattr(RefBase, fn, getattr(U.UNITERModelBase, fn))

    class RefPre(nn.Module):                 # UNITERForPretraining (uniter.py:353-409) for the tasks mlm, itm, mrc
        def __init__(self):
            super().__init__()
            self.loss_configs = {}
            self.mask_probability = c["mask_probability"]
            self.uniter = RefBase()
            self.tasks = ["mlm", "itm", "mrc"] + (["mrfr", "wra"] if all_tasks else [])
            self.heads = nn.ModuleDict({
                "mlm": MLM(OmegaConf.create(dict(type="mlm", vocab_size=V, hidden_size=H))),
                "itm": ITM(OmegaConf.create(dict(type="itm", hidden_size=H))),
                "mrc": MRC(hidden_size=H, label_dim=c["label_dim"])})
            if all_tasks:
                self.heads["mrfr"] = MRFR(self.uniter.img_embeddings.img_linear.weight, hidden_size=H, img_dim=c["img_dim"])   # uniter.py:397-400
                self.heads["wra"] = WRA()
            self.losses = nn.ModuleDict()

    for fn in ("forward", "_process_sample_list_for_pretraining", "_add_image_feat_masked", "_get_img_mask", "_preprocess_mlm",
               "_preprocess_itm", "_preprocess_mrc", "_preprocess_mrfr", "_preprocess_wra", "_get_feature_mask", "_mask_inputs_in_sample_list",
               "_remove_mismatched_captions"):
        setattr(RefPre, fn, getattr(U.UNITERForPretraining, fn))

    ref = RefPre().eval()
    heads = ref.heads["mlm"].cls.predictions
    heads.decoder.bias = heads.bias           # transformers<=4.10 BertLMPredictionHead (the reference's pin)
    # (UNITERForPretraining does not tie the MLM decoder to the word embeddings: the decoder weight is its own parameter)
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()
              if not k.endswith("position_ids") and not k.endswith("embeddings.token_type_ids") and not k.endswith("predictions.decoder.bias")}
    sd = detweights.state_dict(shapes, c["seed"])
    full = {k: torch.from_numpy(v) for k, v in sd.items()}
    full["heads.mlm.cls.predictions.decoder.bias"] = full["heads.mlm.cls.predictions.bias"]
    missing, unexpected = ref.load_state_dict(full, strict=False)
    assert not unexpected, unexpected

    B, T, R, seed = c["B"], c["T"], c["R"], c["seed"]
    u = detweights.uniform
    ids = (u(B * T, seed + 100) * V).astype(np.int64).reshape(B, T)
    mask = np.ones((B, T), dtype=np.int64); mask[1, T // 2:] = 0; mask[2, T - 3:] = 0
    ids[mask == 0] = 0
    pick = (u(B * T, seed + 101).reshape(B, T) < 0.3) & (mask == 1)
    pick[:, 1] = True
    lm = np.where(pick, ids, -1)
    ids_masked = np.where(pick, 3, ids)                                  # 3 = a [MASK] id
    feats = (2.0 * u(B * R * c["img_dim"], seed + 102) - 1.0).astype(np.float32).reshape(B, R, -1)
    xy = u(B * R * 4, seed + 104).astype(np.float32).reshape(B, R, 4)
    x1 = np.minimum(xy[..., 0], xy[..., 2]); x2 = np.maximum(xy[..., 0], xy[..., 2]) + 0.01
    y1 = np.minimum(xy[..., 1], xy[..., 3]); y2 = np.maximum(xy[..., 1], xy[..., 3]) + 0.01
    w, h = x2 - x1, y2 - y1
    pos_feat = np.stack([x1, y1, x2, y2, w, h, w * h], axis=-1).astype(np.float32)
    max_features = np.array([R, R - 2, R - 1], dtype=np.int64)[:B]
    image_valid = (np.arange(R)[None, :] < max_features[:, None]).astype(np.int64)
    attention_mask = np.concatenate([mask, image_valid], axis=-1)
    raw = u(B * R * c["label_dim"], seed + 105).reshape(B, R, -1)
    raw = np.where(raw < 0.5, 0.0, raw) ** 2
    raw[..., 1] += 1e-3
    cls_prob = (raw / raw.sum(-1, keepdims=True)).astype(np.float32)
    is_correct = np.array([1, 0, 1], dtype=np.int64)[:B]

    def sample(task):
        t = torch.from_numpy
        return SampleList(
            input_ids=t(ids.copy()), input_ids_masked=t(ids_masked.copy()), lm_label_ids=t(lm.copy()), input_mask=t(mask.copy()),
            position_ids=torch.arange(0, T, dtype=torch.long).unsqueeze(0), image_feat=t(feats.copy()), img_pos_feat=t(pos_feat.copy()),
            attention_mask=t(attention_mask.copy()), image_mask=t(image_valid.copy()), is_correct=t(is_correct.copy()),
            image_info_0=SampleList(cls_prob=cls_prob.copy()), task=task, dataset_name="coco", dataset_type="train")

    rec = {"in_input_ids": ids, "in_input_ids_masked": ids_masked, "in_lm_label_ids": lm, "in_input_mask": mask, "in_image_feat": feats,
           "in_img_pos_feat": pos_feat, "in_attention_mask": attention_mask, "in_image_mask": image_valid, "in_is_correct": is_correct,
           "in_cls_prob": cls_prob}
    for task in (("mrfr", "wra") if all_tasks else ("mlm", "itm", "mrc")):
        ref.zero_grad()
        np.random.seed(c["seed"] + 7)
        random.seed(c["seed"] + 7)
        sl = sample(task)
        out = ref(sl)
        (key, loss), = out["losses"].items()
        loss.backward()
        rec[task + "_loss"] = np.array(loss.item(), dtype=np.float64)
        rec[task + "_loss_key"] = np.array(key)
        # what the reference's preprocessing handed the encoder and the head (bit-exact targets for the host logic)
        rec[task + "_pre_input_ids"] = sl["input_ids"].numpy()
        rec[task + "_pre_image_feat"] = sl["image_feat"].numpy()
        rec[task + "_pre_image_mask"] = sl["image_mask"].numpy().astype(np.int64)
        if task == "mrc":
            rec["mrc_pre_region_class"] = sl["region_class"].numpy()
            rec["mrc_pre_image_region_mask"] = sl["image_region_mask"].numpy().astype(np.int64)
        if task == "mlm":
            rec["mlm_pre_combined_labels"] = sl["mlm_labels"]["combined_labels"].numpy()
        if task == "mrfr":
            rec["mrfr_pre_region_target"] = sl["mrfr_region_target"].numpy()
            rec["mrfr_pre_region_mask"] = sl["mrfr_region_mask"].numpy().astype(np.int64)
        if task == "wra":
            rec["wra_pre_txt_pad"] = sl["wra_info"]["txt_pad"].numpy().astype(np.int64)
            rec["wra_pre_img_pad"] = sl["wra_info"]["img_pad"].numpy().astype(np.int64)
        names, norms = [], []
        for k, p in ref.named_parameters():
            g = p.grad
            names.append(k)
            norms.append(0.0 if g is None else float(g.double().norm()))
            if g is not None and g.numel() <= 4096 and float(g.abs().max()) > 0:
                rec["grad::%s::%s" % (task, k)] = g.numpy().copy()
        rec[task + "_grad_names"] = np.array(names)
        rec[task + "_grad_norms"] = np.array(norms)
    rec["param_names"] = np.array(list(shapes.keys()))
    rec["param_shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
    rec["state_dict_keys"] = np.array(sorted(k for k in ref.state_dict().keys()
                                             if not k.endswith("position_ids") and not k.endswith("embeddings.token_type_ids")))
    rec["case"] = np.array(repr(c))
    path = os.path.join(HERE, "uniter_pretraining_all.npz" if all_tasks else "uniter_pretraining.npz")
    np.savez_compressed(path, **rec)
    print("uniter_pretraining", {t: float(rec[t + "_loss"]) for t in (("mrfr", "wra") if all_tasks else ("mlm", "itm", "mrc"))}, "->", path,
          os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main(all_tasks="--all-tasks" in sys.argv)
