"""Drop-in hook for a real MMF installation.

    mmf_run config=projects/visual_bert/configs/vqa2/defaults.yaml model=visual_bert dataset=vqa2 \\
            env.user_dir=/path/to/mmf_amd_plugin

MMF imports `env.user_dir` before it builds the config (mmf_cli/run.py:25, mmf/utils/env.py:32-93;
fixture tests/data/user_dir/).  A two-line user dir (`from mmf_amd import plugin; plugin.install()`)
re-registers "visual_bert" and "logit_bce" in MMF's own registry with the HIP-backed classes: a later
`register_model` simply overwrites the dict entry (mmf/common/registry.py:319), and the adapter class
created here derives from MMF's `BaseModel` so the `issubclass` assertion at registry.py:316 holds.
"""


def install():
    from mmf.common.registry import registry as mmf_registry  # noqa: the reference package
    from mmf.models.base_model import BaseModel as MMFBaseModel

    import mmf_amd  # noqa: F401  (fills mmf_amd's registry)
    from mmf_amd.models.visual_bert import VisualBERT as HipVisualBERT
    from mmf_amd.modules.losses import LogitBinaryCrossEntropy as HipLogitBCE

    class VisualBERT(MMFBaseModel):
        """MMF-facing adapter: MMF's BaseModel plumbing (device move, Losses, checkpoints) around the
        HIP-backed network."""

        def __init__(self, config):
            super().__init__(config)
            self.config = config

        @classmethod
        def config_path(cls):
            return HipVisualBERT.config_path()

        @classmethod
        def format_state_key(cls, key):
            return HipVisualBERT.format_state_key(key)

        def build(self):
            inner = HipVisualBERT(self.config)
            inner.build()
            self.model = inner.model
            self._inner = [inner]  # not a sub-module: parameters live under self.model as in the reference

        def get_optimizer_parameters(self, config):
            return self._inner[0].get_optimizer_parameters(config)

        def forward(self, sample_list):
            return HipVisualBERT.forward(self._inner[0], sample_list)

    mmf_registry.register_model("visual_bert")(VisualBERT)
    mmf_registry.register_loss("logit_bce")(HipLogitBCE)
    return VisualBERT
