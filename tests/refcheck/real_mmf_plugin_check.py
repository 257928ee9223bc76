"""Runs in a SUBPROCESS of tests/test_real_mmf_plugin_cpu.py, in the build container only (it needs /root/reference):
the REAL `mmf` package (import shims for the third-party packages this image lacks: tests/golden/refshim.py) with its real
registry, BaseModel, Losses and `mmf.utils.build.build_model`, the REAL model YAMLs of the VQA2 VisualBERT project, and
`mmf_amd.plugin.install()` on top.  Prints one JSON line with what the test asserts on."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import refshim  # noqa: E402

refshim.install()
import torch  # noqa: E402
import yaml  # noqa: E402
from omegaconf import OmegaConf  # noqa: E402  (the shim)

REF = refshim.REF


def merged_model_config():
    """model_config.visual_bert exactly as MMF assembles it for `config=projects/visual_bert/configs/vqa2/defaults.yaml`:
    the model's defaults (mmf/configs/models/visual_bert/defaults.yaml) overlaid by the project file."""
    base = yaml.safe_load(open(os.path.join(REF, "mmf/configs/models/visual_bert/defaults.yaml")))
    proj = yaml.safe_load(open(os.path.join(REF, "projects/visual_bert/configs/vqa2/defaults.yaml")))
    mc = dict(base["model_config"]["visual_bert"])
    mc.update(proj["model_config"]["visual_bert"])
    mc["model"] = "visual_bert"
    return mc, proj


def main():
    out = {}
    registry = refshim.ref_import("mmf.common.registry").registry
    ref_vb = refshim.ref_import("mmf.models.visual_bert")
    build = refshim.ref_import("mmf.utils.build")
    base_model = refshim.ref_import("mmf.models.base_model")
    mc, proj = merged_model_config()
    out["yaml_keys"] = sorted(mc.keys())
    # the reference downloads bert-base-uncased here; no network: same class, same config, random initialisation
    ref_vb.VisualBERTBase.from_pretrained = classmethod(lambda cls, name, config=None, cache_dir=None, **kw: cls(config, **kw))
    registry.register("config", OmegaConf.create({"datasets": "vqa2", "model": "visual_bert", "env": {"cache_dir": "/tmp/mmf_cache", "data_dir": "/tmp/mmf_data"}}))
    ref_cls = registry.get_model_class("visual_bert")
    assert ref_cls is ref_vb.VisualBERT
    ref_model = build.build_model(OmegaConf.create(mc))
    skip = ("position_ids", "embeddings.token_type_ids")
    ref_keys = sorted(k for k in ref_model.state_dict().keys() if not k.endswith(skip))
    ref_shapes = {k: list(v.shape) for k, v in ref_model.state_dict().items() if not k.endswith(skip)}

    # the real register_* decorators import their base classes lazily: resolve those imports through the shim first
    for mod in ("mmf.modules.encoders", "mmf.modules.losses", "mmf.modules.optimizers", "mmf.modules.schedulers",
                "mmf.models.transformers.base"):
        refshim.ref_import(mod)
    from mmf_amd import plugin
    plugin.install()
    hip_cls = registry.get_model_class("visual_bert")
    out["overrides_reference_class"] = hip_cls is not ref_cls
    out["is_real_basemodel_subclass"] = issubclass(hip_cls, base_model.BaseModel)
    hip_model = build.build_model(OmegaConf.create(mc))
    out["built_is_real_basemodel"] = isinstance(hip_model, base_model.BaseModel)
    hip_sd = hip_model.state_dict()
    hip_keys = sorted(k for k in hip_sd.keys() if not k.endswith(skip))
    out["missing_in_hip"] = [k for k in ref_keys if k not in hip_sd]
    out["extra_in_hip"] = [k for k in hip_keys if k not in ref_shapes]
    out["shape_mismatch"] = [k for k in ref_keys if k in hip_sd and list(hip_sd[k].shape) != ref_shapes[k]]
    out["n_keys"] = len(ref_keys)
    # the reference's checkpoint loads into the HIP-backed model (and back)
    res = hip_model.load_state_dict(ref_model.state_dict(), strict=False)
    out["load_unexpected"] = [k for k in res.unexpected_keys if not k.endswith(skip)]
    out["load_missing"] = list(res.missing_keys)
    # MMF's own Losses wrapper got built around the registered HIP loss, keyed as the reference keys it
    out["losses_type"] = type(hip_model.losses).__module__ + "." + type(hip_model.losses).__name__
    # optimizer parameter groups through the model's own hook (mmf/utils/build.py:build_optimizer path uses it)
    full = OmegaConf.create({"model": "visual_bert", "model_config": {"visual_bert": mc}, "optimizer": proj["optimizer"]})
    groups = hip_model.get_optimizer_parameters(full)
    out["optimizer_groups"] = [[len(g["params"]), g.get("weight_decay")] for g in groups]
    # train() / eval() reach the HIP-backed network behind the adapter
    hip_model.eval()
    out["eval_propagates"] = (not hip_model._inner[0].training) and (not hip_model.model.training)
    hip_model.train()
    out["train_propagates"] = hip_model._inner[0].training and hip_model.model.training
    # the adapter's forward under the REAL BaseModel.__call__ (base_model.py:305-337: device move, Mapping assertion, MMF's own `Losses` applied to the
    # model output) with the REAL SampleList — on this CPU-only box every kernel launch is replaced by its extent / dtype checker
    # (tests/native_stub.py), so this proves the plumbing, not the numbers (those are the `-m gpu` tests' job)
    try:
        from tests import native_stub
        sample_mod = refshim.ref_import("mmf.common.sample")
        B, T, R = 2, 128, 100
        sl = sample_mod.SampleList()
        sl.add_field("input_ids", torch.randint(1, 30000, (B, T)))
        sl.add_field("input_mask", torch.ones(B, T, dtype=torch.long))
        sl.add_field("segment_ids", torch.zeros(B, T, dtype=torch.long))
        sl.add_field("image_feature_0", torch.rand(B, R, 2048))
        sl.add_field("targets", torch.zeros(B, 3129))
        sl.dataset_name = "vqa2"
        sl.dataset_type = "train"
        hip_model.train()
        with native_stub.installed() as calls:
            res = hip_model(sl)
            loss = sum(v.sum() for v in res["losses"].values())
            loss.backward()
        out["call_scores_shape"] = list(res["scores"].shape)
        out["call_loss_keys"] = sorted(res["losses"].keys())
        out["call_launched"] = sorted({c[0] for c in calls})[:40]
        out["call_grads"] = sum(1 for p in hip_model.parameters() if p.grad is not None)
    except Exception as e:
        import traceback
        out["call_error"] = "%s: %s | %s" % (type(e).__name__, e, traceback.format_exc()[-800:])
    out["optimizer_type"] = proj["optimizer"]["type"]
    out["registered_optimizer_is_hip"] = registry.get_optimizer_class("adam_w").__module__.startswith("mmf_amd")
    out["registered_scheduler_is_hip"] = registry.get_scheduler_class("warmup_linear").__module__.startswith("mmf_amd")
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
