"""Runs in a SUBPROCESS of tests/test_real_mmf_plugin_cpu.py, in the build container only (it needs /root/reference): the OTHER five
adapters of `mmf_amd.plugin` — vilbert, mmbt, uniter, m4c, mmf_transformer — against the REAL `mmf` package, like
real_mmf_plugin_check.py does for visual_bert:

  1. the reference model is built by the real `mmf.utils.build.build_model` from the model's real YAML files (model defaults overlaid by the
     project file) and its state-dict keys / shapes are recorded;
  2. `plugin.install()` re-registers the HIP-backed adapters in the real registry;
  3. the SAME config goes through the real `build_model` again: the adapter must be a real `BaseModel`, with the reference's state-dict
     keys and shapes, and the reference's checkpoint must load into it.

What is replaced, and only that: network access (`from_pretrained` of HF configs / models constructs the same class from `BertConfig()`, whose
defaults ARE bert-base-uncased), `replace_with_jit` (patches methods the installed transformers no longer has; none of these models calls the
patched classes), the detectron fc7 pickles (zero arrays of the real shape under a temp data dir), OmegaConf interpolations resolved by hand,
and — MMF Transformer only — the raw-image `resnet152` encoder of the image modality (torchvision; SURVEY §2 out of scope) set to `identity`.
Prints one JSON line."""
import json
import os
import pickle
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import transformers.models.auto.configuration_auto as _ca  # noqa: E402
import transformers.models.auto.modeling_auto as _ma  # noqa: E402

sys.modules["transformers.configuration_auto"] = _ca       # (the reference's pin imports the pre-4.x module paths, encoders.py:32-33)
sys.modules["transformers.modeling_auto"] = _ma
import refshim  # noqa: E402

refshim.install()
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import transformers  # noqa: E402
import yaml  # noqa: E402
from omegaconf import OmegaConf  # noqa: E402  (the shim)
from transformers import BertConfig  # noqa: E402
from transformers.modeling_utils import PreTrainedModel  # noqa: E402

REF = refshim.REF
DATA = "/tmp/mmf_refcheck_data"
SKIP = ("position_ids", "embeddings.token_type_ids")


def overlay(a, b):
    for k, v in b.items():
        if k in a and isinstance(a[k], dict) and isinstance(v, dict):
            overlay(a[k], v)
        else:
            a[k] = v
    return a


def fix_floats(v):      # PyYAML reads `1e-12` as a string (YAML 1.1 wants a dot); OmegaConf reads a float
    if isinstance(v, dict):
        return {k: fix_floats(x) for k, x in v.items()}
    if isinstance(v, list):
        return [fix_floats(x) for x in v]
    if isinstance(v, str) and re.fullmatch(r"[+-]?\d+(\.\d*)?[eE][+-]?\d+", v):
        return float(v)
    return v


def model_yaml(key, *files):
    mc = {}
    for f in files:
        y = yaml.safe_load(open(os.path.join(REF, f)))
        overlay(mc, (y.get("model_config") or {}).get(key, {}) or {})
    mc = fix_floats(mc)
    return mc


def configs():
    out = {}
    out["vilbert"] = ("vqa2", model_yaml("vilbert", "mmf/configs/models/vilbert/defaults.yaml", "projects/vilbert/configs/vqa2/defaults.yaml"))
    mc = model_yaml("mmbt", "mmf/configs/models/mmbt/defaults.yaml", "mmf/configs/models/mmbt/classification.yaml",
                    "mmf/configs/models/mmbt/with_features.yaml", "projects/hateful_memes/configs/mmbt/defaults.yaml")
    mc["model_data_dir"] = DATA
    mc["modal_encoder"]["params"]["model_data_dir"] = DATA          # ${model_config.mmbt.model_data_dir}
    out["mmbt"] = ("hateful_memes", mc)
    out["uniter"] = ("vqa2", model_yaml("uniter", "mmf/configs/models/uniter/defaults.yaml"))
    mc = model_yaml("m4c", "mmf/configs/models/m4c/defaults.yaml", "projects/m4c/configs/textvqa/defaults.yaml")
    mc["model_data_dir"] = DATA                                       # ${env.data_dir}
    out["m4c"] = ("textvqa", mc)
    mc = model_yaml("mmf_transformer", "mmf/configs/models/mmf_transformer/defaults.yaml", "projects/hateful_memes/configs/mmf_transformer/defaults.yaml")
    for h in mc["heads"]:
        if isinstance(h.get("num_labels"), str):
            h["num_labels"] = mc["num_labels"]                         # ${model_config.mmf_transformer.num_labels}
    for mod in mc["modalities"]:
        if mod["type"] == "image":
            mod["encoder"] = {"type": "identity", "params": {}}        # raw-image resnet152 (torchvision): out of scope, features in
    out["mmf_transformer"] = ("hateful_memes", mc)
    for name, (_, mc) in out.items():
        mc["model"] = name
        resolve(mc, name, mc)
    return out


def resolve(node, key, root):
    """OmegaConf interpolations the shim's mini-OmegaConf does not resolve: `${model_config.<key>.a.b}` from the model config itself,
    `${env.data_dir}`."""
    items = node.items() if isinstance(node, dict) else enumerate(node)
    for k, v in list(items):
        if isinstance(v, (dict, list)):
            resolve(v, key, root)
        elif isinstance(v, str) and v.startswith("${") and v.endswith("}"):
            path = v[2:-1].split(".")
            if path[:2] == ["model_config", key]:
                cur = root
                for part in path[2:]:
                    cur = cur[part]
                node[k] = cur
            elif path == ["env", "data_dir"]:
                node[k] = DATA


def environment():
    def _select(cfg, key, default=None):
        cur = cfg
        for part in key.split("."):
            if isinstance(cur, dict) and part in cur:
                cur = cur[part]
            else:
                return default
        return cur
    OmegaConf.select = staticmethod(_select)
    os.makedirs(os.path.join(DATA, "models/detectron.defaults"), exist_ok=True)
    for fn, arr in (("fc7_w.pkl", np.zeros((2048, 2048), np.float32)), ("fc7_b.pkl", np.zeros((2048,), np.float32))):
        path = os.path.join(DATA, "models/detectron.defaults", fn)
        if not os.path.exists(path):
            pickle.dump(arr, open(path, "wb"))

    def cfg_from_pretrained(name, **kw):
        kw = {k: v for k, v in kw.items() if k not in ("cache_dir", "force_download", "local_files_only", "revision", "return_unused_kwargs")}
        return BertConfig(**kw)          # BertConfig() == bert-base-uncased
    BertConfig.from_pretrained = classmethod(lambda cls, name, *a, **kw: cfg_from_pretrained(name, **kw))
    transformers.AutoConfig.from_pretrained = staticmethod(lambda name, *a, **kw: cfg_from_pretrained(name, **kw))
    bert_model = transformers.BertModel
    transformers.AutoModel.from_pretrained = staticmethod(lambda name, *a, config=None, **kw: bert_model(config if config is not None else BertConfig()))

    def model_from_pretrained(cls, name, *a, config=None, cache_dir=None, **kw):
        kw = {k: v for k, v in kw.items() if k not in ("force_download", "local_files_only", "revision")}
        return cls(config if config is not None else BertConfig(), *a, **kw)
    PreTrainedModel.from_pretrained = classmethod(model_from_pretrained)

    class PathManager:
        open = staticmethod(open); exists = staticmethod(os.path.exists); isfile = staticmethod(os.path.isfile); isdir = staticmethod(os.path.isdir)
        mkdirs = staticmethod(lambda p: os.makedirs(p, exist_ok=True)); get_local_path = staticmethod(lambda p, **k: p)
    refshim.ref_import("mmf.utils.file_io").PathManager = PathManager
    refshim.ref_import("mmf.modules.hf_layers").replace_with_jit = lambda: None
    for mod in ("mmf.modules.encoders", "mmf.modules.losses", "mmf.modules.optimizers", "mmf.modules.schedulers", "mmf.models.transformers.base",
                "mmf.models.transformers.backends.huggingface", "mmf.models.transformers.heads.mlp", "mmf.models.transformers.heads.mlm",
                "mmf.models.transformers.heads.itm", "mmf.models.transformers.heads.mrc", "mmf.models.transformers.heads.mrfr",
                "mmf.models.transformers.heads.wra", "mmf.models.vilbert", "mmf.models.mmbt", "mmf.models.uniter", "mmf.models.m4c",
                "mmf.models.mmf_transformer"):
        m = refshim.ref_import(mod)
        if "replace_with_jit" in vars(m):
            m.replace_with_jit = lambda: None
        if "PathManager" in vars(m):
            m.PathManager = PathManager
    vb = refshim.ref_import("mmf.models.vilbert")
    vb.ViLBERTBase.from_pretrained = classmethod(lambda cls, name, config=None, cache_dir=None, **kw: cls(config, **kw))


def shapes_of(model):
    return {k: list(v.shape) for k, v in model.state_dict().items() if not k.endswith(SKIP)}


def main():
    environment()
    registry = refshim.ref_import("mmf.common.registry").registry
    build = refshim.ref_import("mmf.utils.build")
    base_model = refshim.ref_import("mmf.models.base_model")
    cfgs = configs()
    registry.register("textvqa_num_final_outputs", 5000 + 50)          # what the TextVQA dataset builder registers (m4c.py:155-163)

    def set_env(dataset, name):
        registry.register("config", OmegaConf.create({"datasets": dataset, "model": name, "env": {"cache_dir": "/tmp/mmf_cache", "data_dir": DATA, "user_dir": ""}}))

    ref_classes, ref_shapes, ref_sd = {}, {}, {}
    for name, (dataset, mc) in cfgs.items():
        set_env(dataset, name)
        ref_classes[name] = registry.get_model_class(name)
        m = build.build_model(OmegaConf.create(mc))
        ref_shapes[name] = shapes_of(m)
        ref_sd[name] = {k: v.clone() for k, v in m.state_dict().items()} if name in ("mmbt", "m4c") else None      # (two of them keep a checkpoint to load)
        del m
    from mmf_amd import plugin
    plugin.install()
    out = {}
    for name, (dataset, mc) in cfgs.items():
        set_env(dataset, name)
        r = {}
        cls = registry.get_model_class(name)
        r["overrides_reference_class"] = cls is not ref_classes[name]
        r["is_real_basemodel_subclass"] = issubclass(cls, base_model.BaseModel)
        try:
            m = build.build_model(OmegaConf.create(mc))
        except Exception as e:          # reported, not raised: one JSON line for the whole run
            r["build_error"] = "%s: %s" % (type(e).__name__, e)
            out[name] = r
            continue
        r["built_is_real_basemodel"] = isinstance(m, base_model.BaseModel)
        hs, rs = shapes_of(m), ref_shapes[name]
        r["n_keys"] = len(rs)
        r["missing_in_hip"] = sorted(k for k in rs if k not in hs)
        r["extra_in_hip"] = sorted(k for k in hs if k not in rs)
        r["shape_mismatch"] = sorted(k for k in rs if k in hs and hs[k] != rs[k])
        if ref_sd[name] is not None:
            res = m.load_state_dict(ref_sd[name], strict=False)
            r["load_unexpected"] = [k for k in res.unexpected_keys if not k.endswith(SKIP)]
            r["load_missing"] = list(res.missing_keys)
        r["losses_type"] = type(m.losses).__module__ + "." + type(m.losses).__name__ if hasattr(m, "losses") else None
        m.eval()
        r["eval_propagates"] = not m._inner[0].training
        m.train()
        r["train_propagates"] = bool(m._inner[0].training)
        out[name] = r
        del m
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
