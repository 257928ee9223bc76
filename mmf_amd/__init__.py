"""mmf_amd — MI355X (gfx950 / CDNA4) native implementation of MMF's cross-modal transformer fusion
hot path (VisualBERT first), behind MMF's registry / BaseModel / SampleList API.

Importing the package registers the HIP-backed components in `mmf_amd.common.registry.registry`
the same way `mmf.utils.env.setup_imports` fills MMF's registry.
"""
from mmf_amd.common.registry import registry  # noqa: F401
from mmf_amd.common.sample import Sample, SampleList  # noqa: F401
from mmf_amd.fp32_path import fp32_inference  # noqa: F401
from mmf_amd.fp32_train import fp32_training  # noqa: F401
from mmf_amd.modules import losses as _losses  # noqa: F401
from mmf_amd.modules import optimizers as _optimizers  # noqa: F401
from mmf_amd.modules import schedulers as _schedulers  # noqa: F401
from mmf_amd.models import visual_bert as _visual_bert  # noqa: F401
from mmf_amd.models import mmbt as _mmbt  # noqa: F401
from mmf_amd.models import mmf_transformer as _mmft  # noqa: F401
from mmf_amd.models import vilbert as _vilbert  # noqa: F401
from mmf_amd.models import uniter as _uniter  # noqa: F401
from mmf_amd.models import m4c as _m4c  # noqa: F401

__version__ = "0.1.0"
