"""The drop-in boundary on the host side (no GPU needed): registry, BaseModel contract, SampleList
semantics and state-dict key compatibility with the reference."""
import pytest
import torch

import mmf_amd
from mmf_amd.common.registry import registry
from mmf_amd.common.sample import Sample, SampleList
from mmf_amd.models.base_model import BaseModel
from oracle import visual_bert_oracle as O
from tests.golden_utils import load_case
from tests.model_utils import model_config
from mmf_amd.utils.build import build_model


def test_components_are_registered_like_mmf_components():
    assert registry.get_model_class("visual_bert") is mmf_amd.models.visual_bert.VisualBERT
    assert registry.get_loss_class("logit_bce") is mmf_amd.modules.losses.LogitBinaryCrossEntropy
    assert issubclass(registry.get_model_class("visual_bert"), BaseModel)
    assert registry.get_model_class("does_not_exist") is None
    with pytest.raises(AssertionError):
        registry.register_model("bad")(object)  # all models must inherit BaseModel (registry.py:316)


def test_registry_key_value_store():
    registry.register("a.b.c", 3)
    assert registry.get("a.b.c") == 3 and registry.get("a.b") == {"c": 3}
    assert registry.get("a.x", default=7) == 7
    registry.unregister("a")
    assert registry.get("a") is None


def test_state_dict_keys_equal_the_reference_parameter_names():
    z, case, cfg, sd, sample = load_case("small64")
    model = build_model(model_config(cfg))
    ours = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    ref = {"model." + k: tuple(s) for k, s in O.parameter_shapes(cfg).items()}
    assert ours == ref
    # and those are the names the real reference model exposed when the fixture was generated
    assert set(str(n) for n in z["param_names"]) == set(ours)


def test_legacy_checkpoint_keys_are_reformatted():
    cls = registry.get_model_class("visual_bert")
    assert cls.format_state_key("bert.bert.encoder.layer.0.output.dense.weight") == "model.bert.encoder.layer.0.output.dense.weight"
    assert cls.format_state_key("bert.classifier.1.bias") == "model.classifier.1.bias"
    z, case, cfg, sd, sample = load_case("tiny")
    cfg = dict(cfg, hidden_size=128, num_attention_heads=2)  # head_dim must be 64 for the HIP kernels
    model = build_model(model_config(cfg))
    legacy = {k.replace("model.bert", "bert.bert").replace("model.classifier", "bert.classifier"): v
              for k, v in model.state_dict().items()}
    model.load_state_dict(legacy, strict=True)


def test_optimizer_parameter_groups_follow_the_bert_recipe():
    z, case, cfg, sd, sample = load_case("small64")
    mc = model_config(cfg)
    model = build_model(mc)
    from mmf_amd.utils.configuration import Config
    full = Config(model="visual_bert", optimizer=dict(params=dict(lr=5e-5)), model_config=dict(visual_bert=mc))
    groups = model.get_optimizer_parameters(full)
    assert [g["weight_decay"] for g in groups] == [0.01, 0.0]
    names = {id(p): n for n, p in model.model.named_parameters()}
    for p in groups[1]["params"]:
        assert "bias" in names[id(p)] or "LayerNorm" in names[id(p)]
    assert sum(len(g["params"]) for g in groups) == len(names)


def test_sample_list_semantics():
    s1 = Sample({"x": torch.ones(3), "info": {"n": torch.tensor(1)}, "name": "a"})
    s2 = Sample({"x": torch.zeros(3), "info": {"n": torch.tensor(2)}, "name": "b"})
    sl = SampleList([s1, s2])
    assert sl.x.shape == (2, 3) and sl.info.n.tolist() == [1, 2] and sl.name == ["a", "b"]
    assert sl.get_batch_size() == 2 and set(sl.fields()) == {"x", "info", "name"}
    sl.add_field("y", torch.zeros(2, 5))
    with pytest.raises(AssertionError):
        sl.add_field("z", torch.zeros(3, 5))
    with pytest.raises(AttributeError):
        sl.missing
    moved = sl.to("cpu")
    assert moved.x.device.type == "cpu" and moved is not sl
    with pytest.raises(TypeError):
        sl.to(3)


def test_model_refuses_to_run_without_hbm_tensors():
    """No silent CPU fallback: the HIP path raises when handed host tensors."""
    z, case, cfg, sd, sample = load_case("small64")
    model = build_model(model_config(cfg))
    from mmf_amd._native import NativeLibraryError
    with pytest.raises((NativeLibraryError, RuntimeError)):
        model(SampleList(sample))


def test_warmup_linear_schedule_matches_transformers():
    import transformers
    from mmf_amd.common.registry import registry as reg
    w = torch.nn.Parameter(torch.zeros(1))
    o1 = torch.optim.SGD([w], lr=5e-5); o2 = torch.optim.SGD([w], lr=5e-5)
    s1 = reg.get_scheduler_class("warmup_linear")(o1, num_warmup_steps=6, num_training_steps=60)
    s2 = transformers.get_linear_schedule_with_warmup(o2, num_warmup_steps=6, num_training_steps=60)
    for _ in range(70):
        assert abs(o1.param_groups[0]["lr"] - o2.param_groups[0]["lr"]) < 1e-12
        o1.step(); o2.step(); s1.step(); s2.step()


def test_adam_w_is_registered():
    from mmf_amd.common.registry import registry as reg
    assert reg.get_optimizer_class("adam_w").__name__ == "AdamW"


def test_nlvr2_head_parameter_tree_matches_the_reference():
    from tests.golden_utils import load_nlvr2_case
    from tests.model_utils import build_visual_bert
    z, case, cfg, sd, sample = load_nlvr2_case()
    model = build_visual_bert(cfg, sd, device="cpu", training_head_type="nlvr2", losses=[dict(type="cross_entropy")])
    ours = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    ref = {str(n): tuple(int(x) for x in str(s).split(",")) for n, s in zip(z["param_names"], z["param_shapes"])}
    assert ours == ref
    assert ours["model.classifier.0.dense.weight"] == (2 * cfg["hidden_size"], 2 * cfg["hidden_size"])


def test_optional_base_model_hooks_follow_the_reference():
    """base_model.py:129-136 (`_run_format_state_key`, in place), :299-303 (`_ensure_sample_list`), :339-351 (`load_requirements`,
    `format_for_prediction`); `build_model` calls `load_requirements` before `build()` (utils/build.py:143-144)."""
    from mmf_amd.common.sample import SampleList
    z, case, cfg, sd, sample = load_case("small64")
    model = build_model(model_config(cfg))
    legacy = {"bert.bert.embeddings.word_embeddings.weight": 1, "bert.classifier.1.weight": 2, "untouched": 3}
    model._run_format_state_key(legacy)
    assert set(legacy) == {model.format_state_key(k) for k in ("bert.bert.embeddings.word_embeddings.weight", "bert.classifier.1.weight", "untouched")}
    assert "untouched" in legacy and "bert.bert.embeddings.word_embeddings.weight" not in legacy
    assert model.format_for_prediction([1, 2], report=None) == [1, 2]
    sl = model._ensure_sample_list({"a": torch.zeros(2)})
    assert isinstance(sl, SampleList) and model._ensure_sample_list(sl) is sl
    with pytest.raises(RuntimeError, match="zoo_requirements"):
        build_model(model_config(cfg, zoo_requirements=["visual_bert.pretrained.coco"]))
    with pytest.raises(RuntimeError, match="zoo_requirements"):
        build_model(model_config(cfg, zoo_requirements="visual_bert.pretrained.coco"))
