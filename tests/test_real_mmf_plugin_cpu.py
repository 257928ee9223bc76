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
