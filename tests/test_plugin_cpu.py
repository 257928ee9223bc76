"""`mmf_amd.plugin.install()` against a stand-in for the MMF package (the real one is not installed here and does not
exist on the GPU box): every HIP-backed component lands in MMF's registry, the model adapters derive from MMF's own
BaseModel (mmf/common/registry.py:316) and keep the reference's parameter tree."""
import sys
import types

import pytest
from torch import nn

from tests.golden_utils import load_m4c_case, load_mmft_case, load_vilbert_case
from tests.model_utils import build_m4c, m4c_model_config, mmft_model_config, vilbert_model_config


@pytest.fixture
def fake_mmf():
    class Registry:
        store = {}
        state = {}

        def get(self, name, default=None, no_warning=False):
            return self.state.get(name, default)

        def __getattr__(self, name):
            if name.startswith("register_"):
                kind = name[len("register_"):]

                def reg(key):
                    def wrap(obj):
                        if kind == "model":
                            assert issubclass(obj, BaseModel), "All models must inherit BaseModel class"
                        self.store[(kind, key)] = obj
                        return obj
                    return wrap
                return reg
            raise AttributeError(name)

    class BaseModel(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.config = config

    mods = {"mmf": types.ModuleType("mmf"), "mmf.common": types.ModuleType("mmf.common"),
            "mmf.common.registry": types.ModuleType("mmf.common.registry"), "mmf.models": types.ModuleType("mmf.models"),
            "mmf.models.base_model": types.ModuleType("mmf.models.base_model")}
    mods["mmf.common.registry"].registry = Registry()
    mods["mmf.models.base_model"].BaseModel = BaseModel
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    yield mods["mmf.common.registry"].registry, BaseModel
    from mmf_amd.common.registry import registry as hip_registry
    type(hip_registry).fallback = None
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def test_install_registers_everything_and_keeps_the_parameter_tree(fake_mmf):
    registry, BaseModel = fake_mmf
    from mmf_amd import plugin
    from mmf_amd.utils.build import build_model
    adapters = plugin.install()
    assert set(adapters) == {"visual_bert", "mmbt", "vilbert", "uniter", "m4c", "mmft", "mmf_transformer"}
    for key in (("loss", "logit_bce"), ("loss", "cross_entropy"), ("optimizer", "adam_w"), ("scheduler", "warmup_linear"),
                ("transformer_backend", "huggingface"), ("transformer_head", "mlp"), ("model", "vilbert"),
                ("loss", "m4c_decoding_bce_with_mask"), ("encoder", "finetune_faster_rcnn_fpn_fc7"), ("model", "m4c")):
        assert key in registry.store, key
    z, case, cfg, sd, sample = load_vilbert_case()
    mc = vilbert_model_config(cfg)
    m = registry.store[("model", "vilbert")](mc)
    assert isinstance(m, BaseModel)
    m.build()
    assert set(m.state_dict().keys()) == set(build_model(mc).state_dict().keys())
    z, case, cfg, sd, sample = load_mmft_case()
    mc = mmft_model_config(cfg)
    m = registry.store[("model", "mmft")](mc)
    m.build()
    assert set(m.state_dict().keys()) == set(build_model(mc).state_dict().keys())
    assert type(m).format_state_key("classifier.2.weight") == "heads.0.classifier.2.weight"
    # M4C: the adapter exposes the sub-modules M4C.build() creates, under the reference's names
    z, case, cfg, sd, sample = load_m4c_case()
    ref_keys = set(build_m4c(cfg, None, device="cpu").state_dict().keys())
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = registry.store[("model", "m4c")](m4c_model_config(cfg))
        m.build()
    assert isinstance(m, BaseModel) and set(m.state_dict().keys()) == ref_keys == {str(n) for n in z["param_names"]}
    # key/value state M4C reads at build time is found in MMF's registry when mmf_amd's own does not hold it
    from mmf_amd.common.registry import registry as hip_registry
    registry.state["someset_num_final_outputs"] = 4242
    assert hip_registry.get("someset_num_final_outputs") == 4242 and hip_registry.get("absent_key", "dflt") == "dflt"


def test_adapter_train_eval_reach_the_hip_network(fake_mmf):
    """The HIP-backed network sits behind the adapter outside the module tree (its parameters are exposed through the
    reference's child names), so train() / eval() are forwarded by hand: M4C picks greedy decoding from `self.training`."""
    registry, BaseModel = fake_mmf
    from mmf_amd import plugin
    plugin.install()
    z, case, cfg, sd, sample = load_m4c_case()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = registry.store[("model", "m4c")](m4c_model_config(cfg))
        m.build()
    inner = m._inner[0]
    m.eval()
    assert not m.training and not inner.training and not inner.mmt.training
    m.train()
    assert m.training and inner.training
    # the branch M4C.forward takes follows the adapter's mode: a dry run counts multimodal-transformer passes (teacher
    # forcing: one) and greedy-decoding calls (eval: the K|V-cached loop, which does not go through mmt.forward)
    from tests import native_stub
    from mmf_amd.common.sample import SampleList
    calls = {"mmt": 0, "greedy": 0}
    orig, orig_dec = type(inner.mmt).forward, type(inner)._decode_incremental

    def counting(self_, *a, **k):
        calls["mmt"] += 1
        return orig(self_, *a, **k)

    def counting_dec(self_, *a, **k):
        calls["greedy"] += 1
        return orig_dec(self_, *a, **k)

    type(inner.mmt).forward = counting
    type(inner)._decode_incremental = counting_dec
    try:
        with native_stub.installed():
            m.train()
            m(SampleList(sample))
            assert calls == {"mmt": 1, "greedy": 0}
            m.eval(); calls.update(mmt=0, greedy=0)
            import torch
            with torch.no_grad():
                m(SampleList(sample))
            assert calls == {"mmt": 0, "greedy": 1}
            inner.config["kv_cached_decode"] = False; calls.update(mmt=0, greedy=0)
            with torch.no_grad():
                m(SampleList(sample))
            assert calls == {"mmt": case["D"], "greedy": 0}
    finally:
        type(inner.mmt).forward = orig
        type(inner)._decode_incremental = orig_dec
