"""Import shims that let the reference (facebookresearch/mmf at /root/reference, read-only) run in the
build container, which lacks omegaconf / torchvision / pytorch-lightning / iopath / torchtext.
Used ONLY by make_golden.py to produce the committed fixtures; nothing at test or run time
imports this (the reference tree does not exist on the GPU box)."""
import sys, types, contextlib
REF = "/root/reference"

class _Any:
    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k): return _Any()
    def __getattr__(self, n):
        if n.startswith("__"): raise AttributeError(n)
        return _Any()
    def __mro_entries__(self, bases): return (object,)

def _stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); m.__path__ = []
    def ga(n):
        if n.startswith("__"): raise AttributeError(n)
        return _Any()
    m.__getattr__ = ga
    sys.modules[name] = m
    return m

def _fakepkg(name, path):
    m = types.ModuleType(name); m.__path__ = [path]; sys.modules[name] = m; return m

# ---- functional mini-omegaconf -------------------------------------------------
class DictConfig(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v
    def __setitem__(self, k, v): super().__setitem__(k, _wrap(v))
    def __getattr__(self, k):
        if k.startswith("__"): raise AttributeError(k)
        try: return self[k]
        except KeyError: raise AttributeError(k)
    def __setattr__(self, k, v): self[k] = v
    def get(self, k, default=None): return self[k] if k in self else default
class ListConfig(list):
    def __init__(self, l=()): super().__init__(_wrap(x) for x in l)
def _wrap(v):
    if isinstance(v, (DictConfig, ListConfig)): return v
    if isinstance(v, dict): return DictConfig(v)
    if isinstance(v, (list, tuple)): return ListConfig(v)
    return v
def _unwrap(v):
    if isinstance(v, dict): return {k: _unwrap(x) for k, x in v.items()}
    if isinstance(v, list): return [_unwrap(x) for x in v]
    return v
class OmegaConf:
    @staticmethod
    def create(x=None): return _wrap(x if x is not None else {})
    @staticmethod
    def to_container(c, resolve=True): return _unwrap(c)
    @staticmethod
    def merge(*cs):
        out = DictConfig()
        def rec(a, b):
            for k, v in b.items():
                if k in a and isinstance(a[k], dict) and isinstance(v, dict): rec(a[k], v)
                else: a[k] = v
        for c in cs: rec(out, c)
        return out
    @staticmethod
    def register_resolver(*a, **k): pass
    register_new_resolver = register_resolver
    @staticmethod
    def set_struct(*a, **k): pass
    @staticmethod
    def set_readonly(*a, **k): pass
    @staticmethod
    def is_dict(c): return isinstance(c, dict)
    @staticmethod
    def is_list(c): return isinstance(c, list)
@contextlib.contextmanager
def open_dict(c): yield c

def install():
    import torch
    from torch import nn
    om = _stub("omegaconf", OmegaConf=OmegaConf, DictConfig=DictConfig, ListConfig=ListConfig,
               open_dict=open_dict, MISSING="???")
    _stub("omegaconf.errors")
    pl = _stub("pytorch_lightning", LightningModule=nn.Module)
    import transformers.models.bert.modeling_bert as mb
    from transformers.modeling_utils import PreTrainedModel
    # transformers<=4.10 semantics (the reference's pin): init_weights == apply(_init_weights)
    def _old_init_weights(self):
        self.apply(self._init_weights)
    PreTrainedModel.init_weights = _old_init_weights
    # transformers<=4.10 `PreTrainedModel._tie_or_clone_weights` (removed later; visual_bert.py:232 calls it): without
    # `config.torchscript` the output embedding SHARES the input embedding's Parameter
    def _old_tie_or_clone_weights(self, output_embeddings, input_embeddings):
        if getattr(self.config, "torchscript", False):
            output_embeddings.weight = nn.Parameter(input_embeddings.weight.clone())
        else:
            output_embeddings.weight = input_embeddings.weight
        if getattr(output_embeddings, "bias", None) is not None:
            output_embeddings.bias.data = nn.functional.pad(
                output_embeddings.bias.data, (0, output_embeddings.weight.shape[0] - output_embeddings.bias.shape[0]), "constant", 0)
        if hasattr(output_embeddings, "out_features") and hasattr(input_embeddings, "num_embeddings"):
            output_embeddings.out_features = input_embeddings.num_embeddings
    PreTrainedModel._tie_or_clone_weights = _old_tie_or_clone_weights
    sys.modules["transformers.modeling_bert"] = mb
    _fakepkg("mmf", REF + "/mmf")
    for sub in ["common", "modules", "models", "utils", "datasets", "trainers"]:
        _fakepkg("mmf." + sub, REF + "/mmf/" + sub)
    import importlib
    for _ in range(80):
        try:
            import mmf.modules.hf_layers, mmf.modules.embeddings, mmf.modules.losses
            import mmf.models.base_model as bm
            sys.modules["mmf.models"].BaseModel = bm.BaseModel
            import mmf.models.visual_bert
            return
        except ModuleNotFoundError as e:
            _stub(e.name)
    raise RuntimeError("could not import reference")


def ref_import(name, tries=200):
    """Import a reference module, stubbing whatever third-party package is missing on the way."""
    import importlib
    for _ in range(tries):
        try:
            return importlib.import_module(name)
        except ModuleNotFoundError as e:
            if e.name and e.name.startswith("mmf."):
                raise
            _stub(e.name)
    raise RuntimeError("could not import " + name)
