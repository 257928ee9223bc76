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
    out["uniter"] = ("vqa2", model_yaml("uniter", "mmf/configs/models/uniter/defaults.yaml", "projects/uniter/configs/vqa2/defaults.yaml"))      # (the project file holds `losses`: with the bare defaults the reference itself has no loss for vqa2)
    mc = model_yaml("m4c", "mmf/configs/models/m4c/defaults.yaml", "projects/m4c/configs/textvqa/defaults.yaml")
    mc["model_data_dir"] = DATA                                       # ${env.data_dir}
    out["m4c"] = ("textvqa", mc)
    mc = model_yaml("mmf_transformer", "mmf/configs/models/mmf_transformer/defaults.yaml", "projects/mmf_transformer/configs/hateful_memes/defaults.yaml")      # (what projects/hateful_memes/configs/mmf_transformer/defaults.yaml includes)
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


def real_sample(name, B=2):
    """A batch of the REAL `mmf.common.sample.SampleList` holding what the model's dataset hands it (shapes of the BASELINE configurations)."""
    import torch
    sample_mod = refshim.ref_import("mmf.common.sample")
    g = torch.Generator().manual_seed(7)
    sl = sample_mod.SampleList()

    def text(T=128):
        ids = torch.randint(1000, 28000, (B, T), generator=g)      # (UNITER's VQA2 project uses bert-base-cased: 28996 ids)
        ids[:, 0] = 101
        sl.add_field("input_ids", ids)
        sl.add_field("input_mask", torch.ones(B, T, dtype=torch.long))
        sl.add_field("segment_ids", torch.zeros(B, T, dtype=torch.long))

    if name == "vilbert":
        text()
        sl.add_field("image_feature_0", torch.rand(B, 100, 2048, generator=g))
        sl.add_field("image_info_0", {"max_features": torch.full((B,), 100, dtype=torch.long), "bbox": torch.rand(B, 100, 5, generator=g)})
        sl.add_field("targets", torch.zeros(B, 3129))
        sl.dataset_name, sl.dataset_type = "vqa2", "train"
    elif name == "mmbt":
        text()
        sl.add_field("image_feature_0", torch.rand(B, 100, 2048, generator=g))
        sl.add_field("targets", torch.randint(0, 2, (B,), generator=g))
        sl.dataset_name, sl.dataset_type = "hateful_memes", "train"
    elif name == "uniter":
        text()
        xy = torch.rand(B, 100, 4, generator=g) * 0.45
        box = torch.stack([xy[..., 0], xy[..., 1], xy[..., 0] + xy[..., 2] + 0.05, xy[..., 1] + xy[..., 3] + 0.05], dim=-1)
        sl.add_field("image_feature_0", torch.rand(B, 100, 2048, generator=g))
        sl.add_field("image_info_0", {"max_features": torch.full((B,), 100, dtype=torch.long), "bbox": box,
                                      "image_width": torch.full((B,), 640), "image_height": torch.full((B,), 480)})
        sl.add_field("targets", torch.zeros(B, 3129))
        sl.dataset_name, sl.dataset_type = "vqa2", "train"
    elif name == "mmf_transformer":
        text()
        sl.add_field("image", torch.rand(B, 100, 2048, generator=g))
        sl.add_field("image_mask", torch.ones(B, 100, dtype=torch.long))
        sl.add_field("targets", torch.randint(0, 2, (B,), generator=g))
        sl.dataset_name, sl.dataset_type = "hateful_memes", "train"
    elif name == "m4c":
        sl.add_field("text", torch.randint(1000, 30000, (B, 20), generator=g))
        sl.add_field("text_len", torch.randint(5, 21, (B,), generator=g))
        sl.add_field("image_feature_0", torch.rand(B, 100, 2048, generator=g))
        sl.add_field("obj_bbox_coordinates", torch.rand(B, 100, 4, generator=g))
        sl.add_field("image_info_0", {"max_features": torch.full((B,), 100, dtype=torch.long)})
        sl.add_field("context_feature_0", torch.randn(B, 50, 300, generator=g))
        sl.add_field("context_feature_1", torch.rand(B, 50, 604, generator=g))
        sl.add_field("image_feature_1", torch.rand(B, 100, 2048, generator=g))
        sl.add_field("ocr_bbox_coordinates", torch.rand(B, 50, 4, generator=g))
        sl.add_field("context_info_0", {"max_features": torch.randint(5, 51, (B,), generator=g)})
        sl.add_field("order_vectors", torch.zeros(B, 50, 50))
        sl.add_field("train_prev_inds", torch.randint(0, 5050, (B, 12), generator=g))
        sl.add_field("targets", (torch.rand(B, 12, 5050, generator=g) > 0.999).float())
        sl.add_field("train_loss_mask", (torch.rand(B, 12, generator=g) > 0.3).float())
        sl.dataset_name, sl.dataset_type = "textvqa", "train"
    return sl


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
        # the adapter's forward + backward under the REAL BaseModel.__call__ with the REAL SampleList (round 6; round 5 did this for VisualBERT only).
        # Every kernel launch is its extent / dtype checker on this CPU-only box (tests/native_stub.py): plumbing, not numbers.
        try:
            from tests import native_stub
            sl = real_sample(name)
            with native_stub.installed() as calls:
                res = m(sl)
                losses = res["losses"]
                loss = sum(v.sum() for v in losses.values())
                loss.backward()
            r["call_scores_shape"] = list(res["scores"].shape) if "scores" in res else None
            r["call_loss_keys"] = sorted(losses.keys())
            r["call_launched"] = sorted({c[0] for c in calls})[:60]
            r["call_grads"] = sum(1 for p_ in m.parameters() if p_.grad is not None)
            r["call_params"] = sum(1 for p_ in m.parameters() if p_.requires_grad)
        except Exception as e:
            import traceback
            r["call_error"] = "%s: %s | %s" % (type(e).__name__, e, traceback.format_exc()[-1200:])
        out[name] = r
        del m
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
