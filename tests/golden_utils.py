"""Helpers shared by the golden-fixture tests: load a fixture written by tests/golden/make_golden.py
and rebuild its (deterministic) weights and inputs."""
import ast
import os

import numpy as np
import torch

from tests.golden import detweights

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, "visual_bert_%s.npz" % name), allow_pickle=False)
    case = ast.literal_eval(str(z["case"]))
    shapes = {str(n): tuple(int(x) for x in str(s).split(",")) for n, s in zip(z["param_names"], z["param_shapes"])}
    sd = {k: torch.from_numpy(v) for k, v in detweights.state_dict(shapes, case["seed"]).items()}
    cfg = dict(
        vocab_size=case["vocab_size"], hidden_size=case["hidden_size"], num_hidden_layers=case["num_hidden_layers"],
        num_attention_heads=case["num_attention_heads"], intermediate_size=case["intermediate_size"],
        max_position_embeddings=case["max_position_embeddings"], type_vocab_size=2, layer_norm_eps=1e-12,
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, visual_embedding_dim=case["visual_embedding_dim"],
        num_labels=case["num_labels"], pooler_strategy="vqa", initializer_range=0.02)
    sample = {
        "input_ids": torch.from_numpy(z["in_input_ids"]), "input_mask": torch.from_numpy(z["in_input_mask"]),
        "segment_ids": torch.from_numpy(z["in_segment_ids"]), "image_feature_0": torch.from_numpy(z["in_image_feature_0"]),
        "image_info_0": {"max_features": torch.from_numpy(z["in_max_features"])},
        "targets": torch.from_numpy(z["in_targets"]), "dataset_name": "vqa2", "dataset_type": "train",
    }
    if "in_image_text_alignment" in z.files:
        sample["image_text_alignment"] = torch.from_numpy(z["in_image_text_alignment"])
    # the reference model prefixes its parameters with "model." (visual_bert.py:420-422)
    sd = {k[len("model."):] if k.startswith("model.") else k: v for k, v in sd.items()}
    return z, case, cfg, sd, sample


def load_mmbt_case(name="mmbt_small64"):
    z = np.load(os.path.join(GOLDEN_DIR, "%s.npz" % name), allow_pickle=False)
    case = ast.literal_eval(str(z["case"]))
    shapes = {str(n)[len("model."):]: tuple(int(x) for x in str(s).split(",")) for n, s in zip(z["param_names"], z["param_shapes"])}
    sd = {k: torch.from_numpy(v) for k, v in detweights.state_dict(shapes, case["seed"]).items()}
    cfg = dict(
        vocab_size=case["vocab_size"], hidden_size=case["hidden_size"], num_hidden_layers=case["num_hidden_layers"],
        num_attention_heads=case["num_attention_heads"], intermediate_size=case["intermediate_size"],
        max_position_embeddings=case["max_position_embeddings"], type_vocab_size=2, layer_norm_eps=1e-12,
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, modal_hidden_size=case["modal_hidden_size"],
        num_labels=case["num_labels"], use_modal_start_token=True, use_modal_end_token=True, num_segments=2,
        initializer_range=0.02)
    if case.get("is_decoder", False):
        cfg["is_decoder"] = True
    sample = {
        "input_ids": torch.from_numpy(z["in_input_ids"]), "input_mask": torch.from_numpy(z["in_input_mask"]),
        "segment_ids": torch.from_numpy(z["in_segment_ids"]), "image_feature_0": torch.from_numpy(z["in_image_feature_0"]),
        "dataset_name": "hateful_memes", "dataset_type": "train",
    }
    if "in_targets" in z.files:
        sample["targets"] = torch.from_numpy(z["in_targets"])
    return z, case, cfg, sd, sample


def load_mmbt_pretraining_case():
    """`mmbt_pretraining`: MMBT with the masked-LM pretraining head; parameters under the reference's names (no `model.` prefix),
    the tied decoder keys are aliases and not listed."""
    z, case, cfg, sd, sample = load_mmbt_case("mmbt_pretraining")
    sample.pop("targets", None)
    sample["lm_label_ids"] = torch.from_numpy(z["in_lm_label_ids"])
    return z, case, cfg, sd, sample


def load_mmft_case(name="mmft_small64"):
    z = np.load(os.path.join(GOLDEN_DIR, "%s.npz" % name), allow_pickle=False)
    case = ast.literal_eval(str(z["case"]))
    shapes = {str(n): tuple(int(x) for x in str(s).split(",")) for n, s in zip(z["param_names"], z["param_shapes"])}
    sd = {k: torch.from_numpy(v) for k, v in detweights.state_dict(shapes, case["seed"]).items()}
    cfg = dict(
        vocab_size=case["vocab_size"], hidden_size=case["hidden_size"], num_hidden_layers=case["num_hidden_layers"],
        num_attention_heads=case["num_attention_heads"], intermediate_size=case["intermediate_size"],
        max_position_embeddings=case["max_position_embeddings"], type_vocab_size=2, layer_norm_eps=1e-12,
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, pad_token_id=0, num_labels=case["num_labels"],
        head_layer_norm_eps=1e-6, head_dropout_prob=0.1, initializer_range=0.02,
        modalities=[
            dict(type="text", key="text", position_dim=case["max_position_embeddings"], segment_id=0,
                 embedding_dim=case["hidden_size"], layer_norm_eps=1e-12, hidden_dropout_prob=0.1),
            dict(type="image", key="image", embedding_dim=case["embedding_dim"], position_dim=case["max_position_embeddings"],
                 segment_id=1, layer_norm_eps=1e-12, hidden_dropout_prob=0.1),
        ])
    sample = {
        "input_ids": torch.from_numpy(z["in_input_ids"]), "input_mask": torch.from_numpy(z["in_input_mask"]),
        "segment_ids": torch.from_numpy(z["in_segment_ids"]), "image": torch.from_numpy(z["in_image"]),
        "image_mask": torch.from_numpy(z["in_image_mask"]), "targets": torch.from_numpy(z["in_targets"]),
        "dataset_name": "hateful_memes", "dataset_type": "train",
    }
    return z, case, cfg, sd, sample


def load_vilbert_case(name="vilbert_small"):
    """`vilbert_small`: classification head; `vilbert_nlvr2`: two images per sample (img0 / img1), paired head; `vilbert_dyn`:
    `vilbert_small` with `dynamic_attention: true`."""
    z = np.load(os.path.join(GOLDEN_DIR, "%s.npz" % name), allow_pickle=False)
    case = ast.literal_eval(str(z["case"]))
    shapes = {str(n)[len("model."):]: tuple(int(x) for x in str(s).split(",")) for n, s in zip(z["param_names"], z["param_shapes"])}
    sd = {k: torch.from_numpy(v) for k, v in detweights.state_dict({"model." + k: s for k, s in shapes.items()}, case["seed"]).items()}
    sd = {k[len("model."):]: v for k, v in sd.items()}
    cfg = dict(
        vocab_size=case["vocab_size"], hidden_size=case["hidden_size"], num_hidden_layers=case["num_hidden_layers"],
        num_attention_heads=case["num_attention_heads"], intermediate_size=case["intermediate_size"],
        max_position_embeddings=case["max_position_embeddings"], type_vocab_size=2, layer_norm_eps=1e-12,
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, pad_token_id=0, v_feature_size=case["v_feature_size"],
        v_hidden_size=case["v_hidden_size"], v_num_hidden_layers=case["v_num_hidden_layers"],
        v_num_attention_heads=case["v_num_attention_heads"], v_intermediate_size=case["v_intermediate_size"],
        bi_hidden_size=case["bi_hidden_size"], bi_num_attention_heads=case["bi_num_attention_heads"],
        bi_intermediate_size=case["bi_intermediate_size"], v_attention_probs_dropout_prob=0.1, v_hidden_dropout_prob=0.1,
        v_biattention_id=list(case["v_biattention_id"]), t_biattention_id=list(case["t_biattention_id"]), fusion_method="mul",
        num_labels=case["num_labels"], initializer_range=0.02, dynamic_attention=bool(case.get("dynamic_attention", False)),
        fixed_t_layer=int(case.get("fixed_t_layer", 0)), fixed_v_layer=int(case.get("fixed_v_layer", 0)),
        in_batch_pairs=bool(case.get("in_batch_pairs", False)), fast_mode=bool(case.get("fast_mode", False)))
    sample = {
        "input_ids": torch.from_numpy(z["in_input_ids"]), "input_mask": torch.from_numpy(z["in_input_mask"]),
        "segment_ids": torch.from_numpy(z["in_segment_ids"]), "image_feature_0": torch.from_numpy(z["in_image_feature_0"]),
        "image_info_0": {"max_features": torch.from_numpy(z["in_max_features"]), "bbox": torch.from_numpy(z["in_bbox"])},
        "targets": torch.from_numpy(z["in_targets"]), "dataset_name": "vqa2", "dataset_type": "train",
    }
    if case.get("nlvr2", False):
        cfg["training_head_type"] = "nlvr2"
        sample = {
            "input_ids": sample["input_ids"], "input_mask": sample["input_mask"], "segment_ids": sample["segment_ids"],
            "img0": {"image_feature_0": sample["image_feature_0"], "image_info_0": sample["image_info_0"]},
            "img1": {"image_feature_0": torch.from_numpy(z["in_feats1"]),
                     "image_info_0": {"max_features": torch.from_numpy(z["in_max_features1"]), "bbox": torch.from_numpy(z["in_bbox1"])}},
            "targets": torch.from_numpy(z["in_labels"]), "dataset_name": "nlvr2", "dataset_type": "train",
        }
    return z, case, cfg, sd, sample


def load_uniter_case(name="uniter_small64"):
    z = np.load(os.path.join(GOLDEN_DIR, "%s.npz" % name), allow_pickle=False)
    case = ast.literal_eval(str(z["case"]))
    shapes = {str(n): tuple(int(x) for x in str(s).split(",")) for n, s in zip(z["param_names"], z["param_shapes"])}
    sd = {k: torch.from_numpy(v) for k, v in detweights.state_dict(shapes, case["seed"]).items()}
    cfg = dict(
        vocab_size=case["vocab_size"], hidden_size=case["hidden_size"], num_hidden_layers=case["num_hidden_layers"],
        num_attention_heads=case["num_attention_heads"], intermediate_size=case["intermediate_size"],
        max_position_embeddings=case["max_position_embeddings"], type_vocab_size=2, layer_norm_eps=1e-12,
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, pad_token_id=0, img_dim=case["img_dim"], pos_dim=7,
        img_hidden_dropout_prob=0.1, task="vqa2", head_hidden_size=case["head_hidden_size"], head_layer_norm_eps=1e-6,
        head_dropout_prob=0.1, num_labels=case["num_labels"], initializer_range=0.02)
    sample = {
        "input_ids": torch.from_numpy(z["in_input_ids"]), "input_mask": torch.from_numpy(z["in_input_mask"]),
        "segment_ids": torch.zeros_like(torch.from_numpy(z["in_input_ids"])),
        "image_feature_0": torch.from_numpy(z["in_image_feature_0"]),
        "image_info_0": {"max_features": torch.from_numpy(z["in_max_features"]), "bbox": torch.from_numpy(z["in_bbox"]),
                         "image_width": torch.from_numpy(z["in_image_width"]), "image_height": torch.from_numpy(z["in_image_height"])},
        "targets": torch.from_numpy(z["in_targets"]), "dataset_name": "vqa2", "dataset_type": "train",
    }
    return z, case, cfg, sd, sample


def load_bypass_case():
    """`visual_bert_bypass`: `bypass_transformer: true` (text-only encoder + one joint `additional_layer`), default pooler strategy."""
    z = np.load(os.path.join(GOLDEN_DIR, "visual_bert_bypass.npz"), allow_pickle=False)
    zz, case, cfg, sd, sample = load_case("small64")
    case = ast.literal_eval(str(z["case"]))
    shapes = {str(n): tuple(int(x) for x in str(s).split(",")) for n, s in zip(z["param_names"], z["param_shapes"])}
    sd = {k[len("model."):]: torch.from_numpy(v) for k, v in detweights.state_dict(shapes, case["seed"]).items()}
    cfg = dict(cfg, bypass_transformer=True, pooler_strategy="default")
    sample = {
        "input_ids": torch.from_numpy(z["in_input_ids"]), "input_mask": torch.from_numpy(z["in_input_mask"]),
        "segment_ids": torch.from_numpy(z["in_segment_ids"]), "image_feature_0": torch.from_numpy(z["in_image_feature_0"]),
        "image_info_0": {"max_features": torch.from_numpy(z["in_max_features"])},
        "targets": torch.from_numpy(z["in_targets"]), "dataset_name": "vqa2", "dataset_type": "train",
    }
    return z, case, cfg, sd, sample


def load_nlvr2_case():
    z = np.load(os.path.join(GOLDEN_DIR, "visual_bert_nlvr2.npz"), allow_pickle=False)
    case = ast.literal_eval(str(z["case"]))
    shapes = {str(n): tuple(int(x) for x in str(s).split(",")) for n, s in zip(z["param_names"], z["param_shapes"])}
    sd = {k: torch.from_numpy(v) for k, v in detweights.state_dict(shapes, case["seed"]).items()}
    sd = {k[len("model."):] if k.startswith("model.") else k: v for k, v in sd.items()}
    cfg = dict(
        vocab_size=case["vocab_size"], hidden_size=case["hidden_size"], num_hidden_layers=case["num_hidden_layers"],
        num_attention_heads=case["num_attention_heads"], intermediate_size=case["intermediate_size"],
        max_position_embeddings=case["max_position_embeddings"], type_vocab_size=2, layer_norm_eps=1e-12,
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, visual_embedding_dim=case["visual_embedding_dim"],
        num_labels=case["num_labels"], pooler_strategy="default", training_head_type="nlvr2", initializer_range=0.02)
    sample = {
        "input_ids": torch.from_numpy(z["in_input_ids"]), "input_mask": torch.from_numpy(z["in_input_mask"]),
        "segment_ids": torch.from_numpy(z["in_segment_ids"]),
        "img0": {"image_feature_0": torch.from_numpy(z["in_feats0"]), "image_info_0": {"max_features": torch.from_numpy(z["in_max_features0"])}},
        "img1": {"image_feature_0": torch.from_numpy(z["in_feats1"]), "image_info_0": {"max_features": torch.from_numpy(z["in_max_features1"])}},
        "image_feature_0": torch.from_numpy(z["in_feats0"]),
        "targets": torch.from_numpy(z["in_targets"]), "dataset_name": "nlvr2", "dataset_type": "train",
    }
    return z, case, cfg, sd, sample


def load_pretraining_case():
    """`visual_bert_pretraining`: VisualBERT with the masked-LM pretraining head (decoder tied to the word embeddings)."""
    z = np.load(os.path.join(GOLDEN_DIR, "visual_bert_pretraining.npz"), allow_pickle=False)
    case = ast.literal_eval(str(z["case"]))
    shapes = {str(n): tuple(int(x) for x in str(s).split(",")) for n, s in zip(z["param_names"], z["param_shapes"])}
    sd = {k: torch.from_numpy(v) for k, v in detweights.state_dict(shapes, case["seed"]).items()}
    sd = {k[len("model."):] if k.startswith("model.") else k: v for k, v in sd.items()}
    cfg = dict(
        vocab_size=case["vocab_size"], hidden_size=case["hidden_size"], num_hidden_layers=case["num_hidden_layers"],
        num_attention_heads=case["num_attention_heads"], intermediate_size=case["intermediate_size"],
        max_position_embeddings=case["max_position_embeddings"], type_vocab_size=2, layer_norm_eps=1e-12,
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, visual_embedding_dim=case["visual_embedding_dim"],
        num_labels=case["num_labels"], pooler_strategy="default", training_head_type="pretraining", initializer_range=0.02)
    sample = {
        "input_ids": torch.from_numpy(z["in_input_ids"]), "input_mask": torch.from_numpy(z["in_input_mask"]),
        "segment_ids": torch.from_numpy(z["in_segment_ids"]), "image_feature_0": torch.from_numpy(z["in_image_feature_0"]),
        "image_info_0": {"max_features": torch.from_numpy(z["in_max_features"])},
        "lm_label_ids": torch.from_numpy(z["in_lm_label_ids"]), "dataset_name": "coco", "dataset_type": "train",
    }
    return z, case, cfg, sd, sample


def load_vilbert_pretraining_case(visual_target=0):
    """`vilbert_pretraining`: ViLBERT with the pretraining heads (masked LM + masked region classification, visual_target 0);
    `visual_target=1`: `vilbert_pretraining_vt1`, the masked-region REGRESSION form (nn.MSELoss, vilbert.py:1139-1148); `visual_target=2`:
    `vilbert_pretraining_vt2`, the NCE form with sampled negatives (:1158-1227)."""
    z = np.load(os.path.join(GOLDEN_DIR, "vilbert_pretraining.npz" if visual_target == 0 else "vilbert_pretraining_vt%d.npz" % visual_target),
                allow_pickle=False)
    case = ast.literal_eval(str(z["case"]))
    shapes = {str(n): tuple(int(x) for x in str(s).split(",")) for n, s in zip(z["param_names"], z["param_shapes"])}
    sd = {k[len("model."):]: torch.from_numpy(v) for k, v in detweights.state_dict(shapes, case["seed"]).items()}
    cfg = dict(
        vocab_size=case["vocab_size"], hidden_size=case["hidden_size"], num_hidden_layers=case["num_hidden_layers"],
        num_attention_heads=case["num_attention_heads"], intermediate_size=case["intermediate_size"],
        max_position_embeddings=case["max_position_embeddings"], type_vocab_size=2, layer_norm_eps=1e-12,
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, pad_token_id=0, v_feature_size=case["v_feature_size"],
        v_hidden_size=case["v_hidden_size"], v_num_hidden_layers=case["v_num_hidden_layers"],
        v_num_attention_heads=case["v_num_attention_heads"], v_intermediate_size=case["v_intermediate_size"],
        bi_hidden_size=case["bi_hidden_size"], bi_num_attention_heads=case["bi_num_attention_heads"],
        bi_intermediate_size=case["bi_intermediate_size"], v_attention_probs_dropout_prob=0.1, v_hidden_dropout_prob=0.1,
        v_biattention_id=list(case["v_biattention_id"]), t_biattention_id=list(case["t_biattention_id"]), fusion_method="mul",
        num_labels=case["num_labels"], initializer_range=0.02, dynamic_attention=False, training_head_type="pretraining",
        v_target_size=case["v_target_size"], visual_target=visual_target)
    sample = {
        "input_ids": torch.from_numpy(z["in_input_ids"]), "input_mask": torch.from_numpy(z["in_input_mask"]),
        "segment_ids": torch.from_numpy(z["in_segment_ids"]), "image_feature_0": torch.from_numpy(z["in_image_feature_0"]),
        "image_info_0": {"max_features": torch.from_numpy(z["in_max_features"]), "bbox": torch.from_numpy(z["in_bbox"]),
                         "cls_prob": torch.from_numpy(z["in_cls_prob"])},
        "image_labels": torch.from_numpy(z["in_image_labels"]), "lm_label_ids": torch.from_numpy(z["in_lm_label_ids"]),
        "dataset_name": "coco", "dataset_type": "train",
    }
    if visual_target == 2:      # NCE: the reference ran with `num_negative: 10`; its (recorded, deterministic) draws give these flat indices
        cfg["num_negative"] = 10
        sample["_negative_index"] = torch.from_numpy(z["in_negative_index"])
    return z, case, cfg, sd, sample


class recorded_random:
    """Context manager: `Tensor.random_` returns the draws recorded in a `visual_target: 2` fixture, in order (the reference ran with the
    same replacement, tests/golden/make_golden.py::make_vilbert_pretraining)."""

    def __init__(self, z):
        self.draws = [z["in_draw%d" % i] for i in range(3)]

    def __enter__(self):
        self.real = torch.Tensor.random_
        it = iter(self.draws)

        def fake(t, lo=0, hi=None):
            d = next(it)
            assert tuple(t.shape) == d.shape, (tuple(t.shape), d.shape)
            return t.copy_(torch.from_numpy(d).to(t.device))
        torch.Tensor.random_ = fake
        return self

    def __exit__(self, *a):
        torch.Tensor.random_ = self.real


def load_transformer_heads_case():
    """`transformer_heads`: the reference's `mlm` / `itm` heads run stand-alone; returns (z, case, {tag: state dict}, inputs)."""
    z = np.load(os.path.join(GOLDEN_DIR, "transformer_heads.npz"), allow_pickle=False)
    case = ast.literal_eval(str(z["case"]))
    sds = {}
    for tag in ("mlm", "itm", "table", "mrc"):
        shapes = {str(n): tuple(int(x) for x in str(s).split(",")) for n, s in zip(z[tag + "_param_names"], z[tag + "_param_shapes"])}
        sds[tag] = {k[len(tag) + 1:]: torch.from_numpy(v) for k, v in detweights.state_dict(shapes, case["seed"]).items()}
    inputs = {"sequence_output": torch.from_numpy(z["in_sequence_output"]), "labels": torch.from_numpy(z["in_labels"]),
              "is_correct": torch.from_numpy(z["in_is_correct"]), "region_mask": torch.from_numpy(z["in_region_mask"]),
              "region_class": torch.from_numpy(z["in_region_class"])}
    return z, case, sds, inputs


def load_transformer_heads_extra():
    """The MRFR / WRA records of `transformer_heads` (oracles ahead of their HIP implementation): parameters of the MRFR head, the tied
    image-embedding weight, the regression targets and the padding masks of the optimal-transport alignment."""
    z, case, sds, inputs = load_transformer_heads_case()
    for tag in ("mrfr", "img"):
        shapes = {str(n): tuple(int(x) for x in str(s).split(",")) for n, s in zip(z[tag + "_param_names"], z[tag + "_param_shapes"])}
        sds[tag] = {k[len(tag) + 1:]: torch.from_numpy(v) for k, v in detweights.state_dict(shapes, case["seed"]).items()}
    inputs = dict(inputs, mrfr_target=torch.from_numpy(z["in_mrfr_target"]), txt_pad=torch.from_numpy(z["in_txt_pad"]),
                  img_pad=torch.from_numpy(z["in_img_pad"]))
    return z, case, sds, inputs


def load_uniter_pretraining_case():
    """`uniter_pretraining`: the reference's UNITERForPretraining run for the tasks mlm / itm / mrc (tests/golden/make_uniter_pretraining.py).
    Parameters carry the prefix `uniter.` (the registered model holds the pretraining wrapper under that name)."""
    z = np.load(os.path.join(GOLDEN_DIR, "uniter_pretraining.npz"), allow_pickle=False)
    case = ast.literal_eval(str(z["case"]))
    shapes = {str(n): tuple(int(x) for x in str(s).split(",")) for n, s in zip(z["param_names"], z["param_shapes"])}
    sd = {"uniter." + k: torch.from_numpy(v) for k, v in detweights.state_dict(shapes, case["seed"]).items()}
    cfg = dict(
        vocab_size=case["vocab_size"], hidden_size=case["hidden_size"], num_hidden_layers=case["num_hidden_layers"],
        num_attention_heads=case["num_attention_heads"], intermediate_size=case["intermediate_size"],
        max_position_embeddings=case["max_position_embeddings"], type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
        attention_probs_dropout_prob=0.1, pad_token_id=0, img_dim=case["img_dim"], label_dim=case["label_dim"])
    T = z["in_input_ids"].shape[1]
    sample = {
        "input_ids": torch.from_numpy(z["in_input_ids"]), "input_ids_masked": torch.from_numpy(z["in_input_ids_masked"]),
        "lm_label_ids": torch.from_numpy(z["in_lm_label_ids"]), "input_mask": torch.from_numpy(z["in_input_mask"]),
        "position_ids": torch.arange(0, T, dtype=torch.long).unsqueeze(0), "image_feat": torch.from_numpy(z["in_image_feat"]),
        "img_pos_feat": torch.from_numpy(z["in_img_pos_feat"]), "attention_mask": torch.from_numpy(z["in_attention_mask"]),
        "image_mask": torch.from_numpy(z["in_image_mask"]), "is_correct": torch.from_numpy(z["in_is_correct"]),
        "image_info_0": {"cls_prob": torch.from_numpy(z["in_cls_prob"])}, "dataset_name": "coco", "dataset_type": "train",
    }
    return z, case, cfg, sd, sample


def load_uniter_pretraining_all_case():
    """`uniter_pretraining_all`: the reference's UNITERForPretraining built with its DEFAULT task list mlm, itm, mrc, mrfr, wra
    (tests/golden/make_uniter_pretraining.py --all-tasks): the records of the tasks mrfr and wra.  Same inputs as `uniter_pretraining`."""
    z = np.load(os.path.join(GOLDEN_DIR, "uniter_pretraining_all.npz"), allow_pickle=False)
    _, case, cfg, _, sample = load_uniter_pretraining_case()
    shapes = {str(n): tuple(int(x) for x in str(s).split(",")) for n, s in zip(z["param_names"], z["param_shapes"])}
    sd = {"uniter." + k: torch.from_numpy(v) for k, v in detweights.state_dict(shapes, case["seed"]).items()}
    # The MRFR head's `linear_proj_weight` IS the image embedding's weight (uniter.py:397-400): the state dict lists the one tensor under
    # both names, and `load_state_dict` leaves it with the value loaded last (the head's entry).
    sd["uniter.uniter.img_embeddings.img_linear.weight"] = sd["uniter.heads.mrfr.linear_proj_weight"]
    return z, case, cfg, sd, sample


def load_m4c_case(name="m4c_small64"):
    z = np.load(os.path.join(GOLDEN_DIR, "%s.npz" % name), allow_pickle=False)
    case = ast.literal_eval(str(z["case"]))
    shapes = {str(n): tuple(int(x) for x in str(s).split(",")) for n, s in zip(z["param_names"], z["param_shapes"])}
    w = detweights.state_dict(shapes, case["seed"])
    for k in w:                    # nn.LayerNorm gains not named "LayerNorm.weight" (make_golden.py::make_m4c)
        if k.endswith("layer_norm.weight"):
            w[k] = w[k] + 1.0
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    cfg = dict(
        text_hidden_size=case["text_hidden_size"], text_num_hidden_layers=case["text_num_hidden_layers"],
        text_num_attention_heads=case["text_num_attention_heads"], text_intermediate_size=case["text_intermediate_size"],
        vocab_size=case["vocab_size"], max_position_embeddings=case["max_position_embeddings"], type_vocab_size=2,
        hidden_size=case["hidden_size"], num_hidden_layers=case["num_hidden_layers"], num_attention_heads=case["num_attention_heads"],
        intermediate_size=case["intermediate_size"], layer_norm_eps=1e-12, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
        obj_in_dim=case["obj_in_dim"], obj_fc7_dim=case["obj_fc7_dim"], ocr_in_dim=case["ocr_in_dim"], ocr_fc7_dim=case["ocr_fc7_dim"],
        fasttext_dim=300, phoc_dim=604, ocr_max_num=case["N"], obj_dropout_prob=0.1, ocr_dropout_prob=0.1,
        num_choices=case["num_choices"], query_key_size=case["query_key_size"], max_dec_length=100, max_type_num=5, bos_idx=1,
        pad_token_id=0)
    t = lambda k: torch.from_numpy(z["in_" + k])
    sample = {
        "text": t("text"), "text_len": t("text_len"), "image_feature_0": t("image_feature_0"),
        "obj_bbox_coordinates": t("obj_bbox_coordinates"), "image_info_0": {"max_features": t("obj_max_features")},
        "context_feature_0": t("context_feature_0"), "context_feature_1": t("context_feature_1"),
        "image_feature_1": t("image_feature_1"), "ocr_bbox_coordinates": t("ocr_bbox_coordinates"),
        "context_info_0": {"max_features": t("ocr_max_features")}, "order_vectors": t("order_vectors"),
        "train_prev_inds": t("train_prev_inds"), "targets": t("targets"), "train_loss_mask": t("train_loss_mask"),
        "dataset_name": "textvqa", "dataset_type": "train",
    }
    return z, case, cfg, sd, sample
