"""Plug-in bus with the reference's surface (mmf/common/registry.py:34-668).

The MI355X components self-register here exactly as MMF components do in `mmf.common.registry`
(`@registry.register_model("visual_bert")`, `register_loss`, `register_encoder`,
`register_transformer_backend`, key/value `register`/`get`).  When the real MMF package is
importable, `mmf_amd.plugin.install()` mirrors every entry into MMF's own registry so the
reference's `build_model(config)` (mmf/utils/build.py:116-151) constructs the HIP-backed classes —
a later registration overwrites an earlier one there (registry.py:319).
"""

_KINDS = (
    "trainer", "builder", "callback", "metric", "torchmetric", "loss", "pool", "fusion", "model", "processor",
    "optimizer", "scheduler", "transformer_backend", "transformer_head", "test_reporter", "decoder", "encoder",
    "datamodule", "iteration_strategy",
)


class Registry:
    mapping = {("%s_name_mapping" % k): {} for k in _KINDS}
    mapping["state"] = {}

    # -- class registries ---------------------------------------------------------------------
    @classmethod
    def _register(cls, kind, name, check=None):
        def wrap(obj):
            if check is not None:
                check(obj)
            cls.mapping["%s_name_mapping" % kind][name] = obj
            return obj

        return wrap

    @classmethod
    def register_model(cls, name):
        def check(model_cls):
            from mmf_amd.models.base_model import BaseModel

            assert issubclass(model_cls, BaseModel), "All models must inherit BaseModel class"

        return cls._register("model", name, check)

    @classmethod
    def _get(cls, kind, name):
        return cls.mapping["%s_name_mapping" % kind].get(name, None)

    # -- key/value state (registry.py:520-668) ---------------------------------------------------
    @classmethod
    def register(cls, name, obj):
        path = name.split(".")
        cur = cls.mapping["state"]
        for part in path[:-1]:
            cur = cur.setdefault(part, {})
        cur[path[-1]] = obj

    # another registry consulted for key/value state this one does not hold: `mmf_amd.plugin.install()` points it at
    # MMF's own registry, where MMF's dataset builders put e.g. `<dataset>_num_final_outputs` and the answer processor
    # that M4C reads at build time (mmf/models/m4c.py:39,158,172)
    fallback = None

    @classmethod
    def get(cls, name, default=None, no_warning=False):
        value = cls.mapping["state"]
        for part in name.split("."):
            if not isinstance(value, dict):
                value = default
                break
            value = value.get(part, default)
            if value is default:
                break
        if value is default and cls.fallback is not None:
            return cls.fallback.get(name, default, no_warning=True)
        return value

    @classmethod
    def unregister(cls, name):
        return cls.mapping["state"].pop(name, None)


def _make_accessors():
    for kind in _KINDS:
        if kind != "model":
            setattr(Registry, "register_%s" % kind,
                    classmethod(lambda cls, name, _k=kind: cls._register(_k, name)))
        getter = classmethod(lambda cls, name, _k=kind: cls._get(_k, name))
        setattr(Registry, "get_%s_class" % kind, getter)
    # spelling kept from the reference (registry.py:607)
    Registry.get_test_rerporter_class = Registry.get_test_reporter_class
    Registry.register_pooler = Registry.register_pool


_make_accessors()
registry = Registry()
