"""Drop-in hook for a real MMF installation.

    mmf_run config=projects/visual_bert/configs/vqa2/defaults.yaml model=visual_bert dataset=vqa2 \\
            env.user_dir=/path/to/mmf_amd_plugin

MMF imports `env.user_dir` before it builds the config (mmf_cli/run.py:25, mmf/utils/env.py:32-93;
fixture tests/data/user_dir/).  A two-line user dir (`from mmf_amd import plugin; plugin.install()`)
re-registers the HIP-backed components in MMF's own registry: a later `register_*` simply overwrites the dict
entry (mmf/common/registry.py:319), and the model adapters created here derive from MMF's `BaseModel`, so the
`issubclass` assertion at registry.py:316 holds and MMF's `build_model` (mmf/utils/build.py:116-151), `Losses`,
checkpointing and trainer loop run unchanged around the HIP-backed networks.

Registered: models `visual_bert`, `mmbt`, `vilbert`, `uniter`, `m4c`, `mmft` / `mmf_transformer`; losses `logit_bce`,
`cross_entropy`, `m4c_decoding_bce_with_mask`; encoders `finetune_faster_rcnn_fpn_fc7`, `transformer` (the HIP-backed `BertModelJit`), `identity`; optimizer `adam_w`; scheduler `warmup_linear`; transformer backend `huggingface`; transformer heads
`mlp` / `multilayer_mlp`, `mlm`, `itm`, `mrc`.
"""


def _adapter(mmf_base_model, hip_cls, children):
    """MMF-facing adapter: MMF's BaseModel plumbing (device move, Losses, checkpoints) around the HIP-backed network.
    `children` are the sub-module names the reference model owns directly (`model` for VisualBERT / MMBT / ViLBERT,
    `backend` / `encoders` / `heads` for MMF Transformer), so the parameter tree keeps the reference's names."""

    class Adapter(mmf_base_model):
        def __init__(self, config, *args, **kwargs):
            super().__init__(config)
            self.config = config

        @classmethod
        def config_path(cls):
            return hip_cls.config_path()

        @classmethod
        def format_state_key(cls, key):
            return hip_cls.format_state_key(key)

        def build(self):
            inner = hip_cls(self.config)
            inner.build()
            for name in children:
                setattr(self, name, getattr(inner, name))
            self._inner = [inner]  # not a sub-module: the parameters live under `children`, as in the reference

        def init_losses(self):
            from mmf_amd.models.base_model import BaseModel as HipBaseModel
            if hip_cls.init_losses is not HipBaseModel.init_losses:      # e.g. UNITER defers losses to its sub-model
                return hip_cls.init_losses(self._inner[0])
            return super().init_losses()

        def get_optimizer_parameters(self, config):
            return self._inner[0].get_optimizer_parameters(config)

        def train(self, mode=True):
            # the HIP-backed network is deliberately NOT a registered sub-module (its parameters live under `children`), so
            # nn.Module.train() does not reach it: forward it by hand — M4C.forward branches on `self.training` (teacher
            # forcing against greedy decoding, mmf/models/m4c.py:286-305)
            super().train(mode)
            inner = self.__dict__.get("_inner")
            if inner:
                inner[0].train(mode)
            return self

        def forward(self, sample_list):
            self._inner[0].training = self.training
            return hip_cls.forward(self._inner[0], sample_list)

    Adapter.__name__ = Adapter.__qualname__ = hip_cls.__name__
    return Adapter


# the sub-modules M4C.build() creates directly on the model (mmf/models/m4c.py:46-170)
M4C_CHILDREN = ("text_bert", "text_bert_out_linear", "obj_faster_rcnn_fc7", "linear_obj_feat_to_mmt_in", "linear_obj_bbox_to_mmt_in",
                "obj_feat_layer_norm", "obj_bbox_layer_norm", "obj_drop", "ocr_faster_rcnn_fc7", "linear_ocr_feat_to_mmt_in",
                "linear_ocr_bbox_to_mmt_in", "ocr_feat_layer_norm", "ocr_bbox_layer_norm", "ocr_drop", "mmt", "ocr_ptr_net", "classifier")


def install():
    from mmf.common.registry import registry as mmf_registry  # noqa: the reference package
    from mmf.models.base_model import BaseModel as MMFBaseModel

    import mmf_amd  # noqa: F401  (fills mmf_amd's registry)
    from mmf_amd.common.registry import registry as hip_registry
    from mmf_amd.models.m4c import M4C
    from mmf_amd.models.mmbt import MMBT
    from mmf_amd.models.mmf_transformer import MMFTransformer
    from mmf_amd.models.uniter import UNITER
    from mmf_amd.models.vilbert import ViLBERT
    from mmf_amd.models.visual_bert import VisualBERT

    type(hip_registry).fallback = mmf_registry      # key/value state (dataset sizes, processors) lives in MMF's registry
    adapters = {}
    for names, cls, children in ((("visual_bert",), VisualBERT, ("model",)), (("mmbt",), MMBT, ("model",)),
                                 (("vilbert",), ViLBERT, ("model",)), (("uniter",), UNITER, ("uniter",)),
                                 (("m4c",), M4C, M4C_CHILDREN),
                                 (("mmft", "mmf_transformer"), MMFTransformer, ("backend", "encoders", "heads"))):
        adapter = _adapter(MMFBaseModel, cls, children)
        for name in names:
            mmf_registry.register_model(name)(adapter)
            adapters[name] = adapter
    for kind, names in (("loss", ("logit_bce", "cross_entropy", "m4c_decoding_bce_with_mask")), ("encoder", ("finetune_faster_rcnn_fpn_fc7", "transformer", "identity")), ("optimizer", ("adam_w",)), ("scheduler", ("warmup_linear",)),
                        ("transformer_backend", ("huggingface",)), ("transformer_head", ("mlp", "multilayer_mlp", "mlm", "itm", "mrc"))):
        for name in names:
            obj = getattr(hip_registry, "get_%s_class" % kind)(name)
            if kind == "encoder":
                # MMF's registry asserts issubclass(encoder, mmf.modules.encoders.Encoder) (registry.py:443-447)
                try:
                    from mmf.modules.encoders import Encoder as MMFEncoder
                except ImportError:       # a stripped-down MMF without the encoders module
                    MMFEncoder = None
                if MMFEncoder is not None and not issubclass(obj, MMFEncoder):
                    obj = type(obj.__name__, (obj, MMFEncoder), {"__doc__": obj.__doc__})
            getattr(mmf_registry, "register_%s" % kind)(name)(obj)
    return adapters
