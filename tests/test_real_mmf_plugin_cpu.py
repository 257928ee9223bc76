"""`mmf_amd.plugin.install()` against the REAL reference package (build container only: /root/reference does not exist on the
GPU box): the real `mmf.common.registry.register_*` decorators with their issubclass assertions, the real
`mmf.utils.build.build_model`, the real `BaseModel` / `Losses`, and the model config assembled from the real YAML files of the
VQA2 VisualBERT project (projects/visual_bert/configs/vqa2/defaults.yaml over mmf/configs/models/visual_bert/defaults.yaml).
Runs in a subprocess because making the reference importable here patches sys.modules heavily (tests/golden/refshim.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/mmf"), reason="the reference tree only exists in the build container")


@pytest.fixture(scope="module")
def result():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "refcheck", "real_mmf_plugin_check.py")], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-4000:])
    return json.loads(lines[-1][len("RESULT "):])


def test_real_yaml_config_runs_unmodified(result):
    # every key of the real model config was accepted as is (the build did not need an edited copy)
    for k in ("bert_model_name", "training_head_type", "pooler_strategy", "num_labels", "losses", "visual_embedding_dim",
              "special_visual_initialize", "finetune_lr_multiplier", "zerobias"):
        assert k in result["yaml_keys"]


def test_plugin_overrides_the_reference_class_in_the_real_registry(result):
    assert result["overrides_reference_class"] and result["is_real_basemodel_subclass"] and result["built_is_real_basemodel"]
    assert result["losses_type"] == "mmf.modules.losses.Losses"          # MMF's own wrapper, around the registered HIP loss
    assert result["registered_optimizer_is_hip"] and result["registered_scheduler_is_hip"] and result["optimizer_type"] == "adam_w"


def test_state_dict_keys_and_shapes_equal_the_reference_models(result):
    assert result["n_keys"] > 200
    assert result["missing_in_hip"] == [] and result["extra_in_hip"] == [] and result["shape_mismatch"] == []
    assert result["load_unexpected"] == [] and result["load_missing"] == []


def test_optimizer_groups_follow_the_bert_decay_rule(result):
    (n_decay, wd), (n_nodecay, wd0) = result["optimizer_groups"]
    assert wd == 0.01 and wd0 == 0.0 and n_decay > 0 and n_nodecay > n_decay     # biases + LayerNorms outnumber the matrices


def test_train_eval_reach_the_network_behind_the_adapter(result):
    assert result["eval_propagates"] and result["train_propagates"]


def test_adapter_forward_runs_under_the_real_basemodel_call(result):
    """Round 5: the adapter's `forward` under the REAL `BaseModel.__call__` (mmf/models/base_model.py:305-337) with the REAL `SampleList`: MMF's own
    `Losses` keys the loss as the reference does, the backward reaches every trainable parameter but the BertPooler (`pooler_strategy: vqa`).  Every
    kernel launch is its extent / dtype checker on this CPU-only box (tests/native_stub.py): plumbing, not numbers."""
    assert "call_error" not in result, result.get("call_error")
    assert result["call_scores_shape"] == [2, 3129] and result["call_loss_keys"] == ["train/vqa2/logit_bce"]
    assert {"gemm", "attention_fwd", "attention_bwd", "layernorm_fwd", "bce_logits_fwd"} <= set(result["call_launched"])
    assert result["call_grads"] >= 200


@pytest.fixture(scope="module")
def models_result():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "refcheck", "real_mmf_plugin_models_check.py")], capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-4000:])
    return json.loads(lines[-1][len("RESULT "):])


@pytest.mark.parametrize("name", ["vilbert", "mmbt", "uniter", "m4c", "mmf_transformer"])
def test_the_other_adapters_build_through_the_real_build_model(models_result, name):
    """each of the other five adapters of mmf_amd/plugin.py is constructed by the REAL `mmf.utils.build.build_model`
    (mmf/utils/build.py:116-151) from its REAL YAML files (mmf/configs/models/<model>/*.yaml overlaid by projects/*/configs: ViLBERT VQA2, MMBT
    hateful_memes with_features, UNITER defaults, M4C TextVQA, MMF Transformer hateful_memes) and compared with the reference model built the same
    way from the same config: same class hierarchy (a real BaseModel), same state-dict keys, same shapes, the reference checkpoint loads."""
    r = models_result[name]
    assert "build_error" not in r, r.get("build_error")
    assert r["overrides_reference_class"] and r["is_real_basemodel_subclass"] and r["built_is_real_basemodel"]
    assert r["n_keys"] > 100
    assert r["missing_in_hip"] == [] and r["extra_in_hip"] == [] and r["shape_mismatch"] == []
    if "load_unexpected" in r:
        assert r["load_unexpected"] == [] and r["load_missing"] == []
    assert r["eval_propagates"] and r["train_propagates"]
    if name != "uniter":          # (UNITER defers its losses to the per-task heads: uniter.py:640-660)
        assert r["losses_type"] == "mmf.modules.losses.Losses"


@pytest.mark.parametrize("name,scores,loss_key,min_grads", [
    ("vilbert", [2, 3129], "train/vqa2/logit_bce", 500), ("mmbt", [2, 2], "train/hateful_memes/cross_entropy", 209),
    ("uniter", [2, 3129], "train/vqa2/logit_bce", 216), ("m4c", [2, 12, 5050], "train/textvqa/m4c_decoding_bce_with_mask", 151),
    ("mmf_transformer", [2, 2], "train/hateful_memes/cross_entropy", 212)])
def test_the_other_adapters_forward_and_backward_under_the_real_basemodel_call(models_result, name, scores, loss_key, min_grads):
    """Round 6 ("the other five are constructed and key-checked, never called"): each adapter's `forward` under the REAL
    `BaseModel.__call__` (mmf/models/base_model.py:305-337) with a REAL `mmf.common.sample.SampleList` of its BASELINE shapes, built from its real
    YAMLs: MMF's own `Losses` (UNITER: its per-task `MMFLoss`, mmf/models/uniter.py:333-343) key the loss as the reference does, and the backward
    reaches the trainable parameters (the poolers the reference computes and drops, mmf/models/vilbert.py:1322, stay without gradient).  Kernel
    launches are extent / dtype checkers on this CPU-only box (tests/native_stub.py): plumbing, not numbers - those are the `-m gpu` tests' job."""
    r = models_result[name]
    assert "call_error" not in r, r.get("call_error")
    assert r["call_scores_shape"] == scores and r["call_loss_keys"] == [loss_key]
    assert r["call_grads"] >= min_grads and r["call_grads"] >= 0.95 * r["call_params"]
    assert {"gemm", "attention_fwd", "attention_bwd", "layernorm_fwd"} <= set(r["call_launched"]) or len(r["call_launched"]) == 60
