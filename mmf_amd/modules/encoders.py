"""The encoders of mmf/modules/encoders.py that are on the cross-modal transformer paths: the encoder base / factory interface
(:44-64), the image-FEATURE encoders (`ImageFeatureEncoderFactory` :66-113, `IdentityEncoder` :183-198, `FinetuneFasterRcnnFpnFc7`
:116-180), the text encoders (`TextEncoderFactory` :449-479, `TransformerEncoder` :513-585 — the registered `"transformer"` encoder,
here the HIP-backed `BertModelJit`) and `MultiModalEncoderBase` (:588-646), which MMBT's base derives from (mmf/models/mmbt.py:327).
The raw-pixel / video / audio encoders of that file (ResNet152, torchvision, detectron2, FRCNN, ViT, R(2+1)D, ...) are CNN feature
extractors outside SURVEY.md §8's scope: asking a factory for one raises `NotImplementedError`.


`FinetuneFasterRcnnFpnFc7` (encoders.py:117-180): the detector's fc7 layer applied to pre-extracted fc6 region features,
`relu(lc(x))`, kept trainable at a reduced learning rate (m4c.py:104-106).  Same parameter tree (`lc.weight`, `lc.bias`),
same config keys (`in_dim`, `weights_file`, `bias_file`, `model_data_dir`).  The reference downloads the detectron pickles
when they are missing (encoders.py:135-138); there is no network here, so missing files mean a BERT-style random init of
`lc` with `out_dim` (default: `in_dim`) outputs — and a warning."""
import os
import pickle
import warnings
from enum import Enum

import torch
from torch import nn

from mmf_amd import functional as Fn
from mmf_amd import fp32_path as F32P
from mmf_amd import fp32_train as F32T
from mmf_amd.common.registry import registry
from mmf_amd.modules.hf_layers import BertConfig, BertModelJit, Linear
from mmf_amd.utils.configuration import Config, to_container


def _cget(config, key, default=None):
    if config is None:
        return default
    return config.get(key, default) if hasattr(config, "get") else getattr(config, key, default)


class Encoder(nn.Module):
    """encoders.py:44-57.  `Config` is the attribute-dict of mmf_amd.utils.configuration (the reference uses OmegaConf dataclasses)."""

    def __init__(self):
        super().__init__()

    @classmethod
    def from_params(cls, **kwargs):
        return cls(Config(kwargs))


class EncoderFactory(nn.Module):
    """encoders.py:59-64: `config.type` + `config.params` -> `self.module`."""


def _enum_value(t):
    return t.value if isinstance(t, Enum) else t


class ImageFeatureEncoderTypes(Enum):
    default = "default"
    identity = "identity"
    projection = "projection"
    frcnn_fc7 = "finetune_faster_rcnn_fpn_fc7"


class _Identity(nn.Identity):
    pass


class ImageFeatureEncoderFactory(EncoderFactory):
    """encoders.py:79-113: encoders applied to PRE-EXTRACTED region / grid features."""

    def __init__(self, config, *args, **kwargs):
        super().__init__()
        encoder_type = _enum_value(config.type)
        params = _cget(config, "params", None)
        assert params is not None and "in_dim" in params, "ImageFeatureEncoder require 'in_dim' param in config"
        if encoder_type in ("default", "identity"):
            self.module = _Identity()
            self.module.in_dim = params.in_dim
            self.module.out_dim = params.in_dim
        elif encoder_type == "projection":
            if _cget(params, "module", "linear") != "linear":
                raise NotImplementedError("ProjectionEmbedding module=%r: only the linear projection is built" % (params.module,))
            self.module = Linear(params.in_dim, params.out_dim)
            self.module.out_dim = params.out_dim
        elif encoder_type == "finetune_faster_rcnn_fpn_fc7":
            self.module = FinetuneFasterRcnnFpnFc7(params)
        else:
            raise NotImplementedError("Unknown Image Encoder: %s" % encoder_type)
        self.out_dim = self.module.out_dim

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


class ImageEncoderTypes(Enum):
    default = "default"
    identity = "identity"
    torchvision_resnet = "torchvision_resnet"
    resnet152 = "resnet152"
    detectron2_resnet = "detectron2_resnet"


class ImageEncoderFactory(EncoderFactory):
    """encoders.py:209-243.  Only the pass-through types are built: the CNN encoders run ahead of the fusion path (features are
    pre-extracted) and are out of scope."""

    def __init__(self, config, *args, **kwargs):
        super().__init__()
        self._type = _enum_value(config.type)
        params = _cget(config, "params", None) or Config()
        if self._type in ("default", "identity"):
            self.module = _Identity()
            self.module.out_dim = _cget(params, "in_dim", None)
        else:
            raise NotImplementedError(
                "image encoder %r (mmf/modules/encoders.py:246-446) is a CNN feature extractor, out of the fusion path's scope: "
                "feed pre-extracted features (`direct_features_input: true`)" % (self._type,))

    @property
    def out_dim(self):
        return self.module.out_dim

    def forward(self, image):
        return self.module(image)


@registry.register_encoder("identity")
class IdentityEncoder(Encoder):
    """encoders.py:183-198."""

    def __init__(self, config=None):
        super().__init__()
        self.module = nn.Identity()
        self.in_dim = _cget(config, "in_dim", 100)
        self.out_dim = self.in_dim

    def forward(self, x):
        return self.module(x)


# bert-base-uncased / bert-large-uncased: what AutoConfig.from_pretrained(bert_model_name) yields (no network here)
_BERT_BASES = {
    "bert-base-uncased": dict(),
    "bert-large-uncased": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096),
}


@registry.register_encoder("transformer")
class TransformerEncoder(Encoder):
    """encoders.py:513-585: the BERT text encoder behind `text_encoder: {type: transformer, params: ...}` — here `BertModelJit` on the
    HIP kernels.  `params` override the base architecture's config (`_build_encoder_config`, :580-583); `num_segments` re-sizes the
    token-type table the way `_init_segment_embeddings` does (:567-578).  Pretrained weights are not downloaded (no network): the
    module starts from the reference's random init and takes its weights from `load_state_dict` / an MMF checkpoint."""

    def __init__(self, config, *args, **kwargs):
        super().__init__()
        self.config = config
        name = _cget(config, "bert_model_name", "bert-base-uncased") or "bert-base-uncased"
        if not str(name).startswith("bert-"):
            raise NotImplementedError("TransformerEncoder: bert_model_name=%r — only BERT encoders (BertModelJit) are built" % (name,))
        self.module = BertModelJit(self._build_encoder_config(config))
        self.embeddings = self.module.embeddings
        self.original_config = self.config
        self.config = self.module.config
        self._init_segment_embeddings()

    def _init_segment_embeddings(self):
        num_segments = _cget(self.original_config, "num_segments", None)
        if num_segments and hasattr(self.embeddings, "token_type_embeddings"):
            old = self.embeddings.token_type_embeddings.weight
            new_embeds = nn.Embedding(num_segments, self.config.hidden_size)
            new_embeds.weight.data[:2].copy_(old.data[:2])
            for idx in range(2, num_segments - 1):
                new_embeds.weight.data[idx].copy_(old.data.mean(dim=0))
            self.embeddings.token_type_embeddings = new_embeds
            self.config.type_vocab_size = num_segments

    def _build_encoder_config(self, config):
        name = _cget(config, "bert_model_name", "bert-base-uncased") or "bert-base-uncased"
        if name not in _BERT_BASES:
            raise NotImplementedError("bert_model_name=%r: known bases are %s" % (name, sorted(_BERT_BASES)))
        d = dict(_BERT_BASES[name])
        d.update({k: v for k, v in to_container(config).items()
                  if not isinstance(v, (dict, list)) and k not in ("name", "num_segments", "bert_model_name", "random_init")})
        return BertConfig.from_dict(d)

    def forward(self, *args, return_sequence=False, **kwargs):
        output = self.module(*args, **kwargs)          # (sequence_output, pooled_output)
        return output[0] if return_sequence else output[1]


class TextEncoderTypes(Enum):
    identity = "identity"
    transformer = "transformer"
    embedding = "embedding"


class TextEncoderFactory(EncoderFactory):
    """encoders.py:455-479."""

    def __init__(self, config, *args, **kwargs):
        super().__init__()
        self._type = _enum_value(config.type)
        if self._type == "identity":
            self.module = nn.Identity()
        elif self._type == "transformer":
            self._module = TransformerEncoder(_cget(config, "params", None) or Config())
            self.module = self._module.module
        elif self._type == "embedding":
            raise NotImplementedError("TextEmbeddingEncoder (encoders.py:482-510: word-vector / RNN text embeddings) is not on the "
                                      "transformer fusion path")
        else:
            raise NotImplementedError("Unknown Text Encoder %s" % self._type)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def build_text_encoder(config, *args, **kwargs):
    """mmf/utils/build.py:495-503."""
    return TextEncoderFactory(config, *args, **kwargs).module


def build_image_encoder(config, direct_features=False, **kwargs):
    """mmf/utils/build.py:506-514."""
    module = ImageFeatureEncoderFactory(config) if direct_features else ImageEncoderFactory(config)
    return module.module


def build_encoder(config):
    """mmf/utils/build.py:517-545: `{type, params}` or a structured config carrying `name`, resolved through the encoder registry."""
    if "type" in config:
        name, params = _enum_value(config.type), _cget(config, "params", None)
    else:
        name, params = config.name, config
    encoder_cls = registry.get_encoder_class(name)
    if encoder_cls is None:
        raise NotImplementedError("no encoder registered under %r" % (name,))
    return encoder_cls(params)


class MultiModalEncoderBase(Encoder):
    """encoders.py:588-646: owns a text encoder and a modal encoder built from `config.text_encoder` / `config.modal_encoder`."""

    def __init__(self, config, *args, **kwargs):
        super().__init__()
        self.config = config
        self._modal_encoder_config = _cget(self.config, "modal_encoder", None)
        self._is_direct_features_input = _cget(self.config, "direct_features_input", False)
        self.build()
        self.modal_hidden_size = _cget(self.config, "modal_hidden_size", None)
        self.text_hidden_size = _cget(self.config, "text_hidden_size", None)

    def build(self):
        encoders = self._build_encoders(self.config)
        self.text_encoder, self.modal_encoder = encoders[0], encoders[1]
        self._encoder_config = None
        if self.text_encoder:
            self._encoder_config = self.text_encoder.config

    @property
    def encoder_config(self):
        return self._encoder_config

    def _build_encoders(self, config):
        text_encoder = None
        if _cget(config, "text_encoder", None):
            text_encoder = build_text_encoder(config.text_encoder)
        modal_encoder = None
        if _cget(config, "modal_encoder", None):
            modal_encoder = self._build_modal_encoder(config.modal_encoder)
        return (text_encoder, modal_encoder)

    def _build_modal_encoder(self, config):
        return build_image_encoder(config, direct_features=self._is_direct_features_input)


@registry.register_encoder("finetune_faster_rcnn_fpn_fc7")
class FinetuneFasterRcnnFpnFc7(Encoder):
    def __init__(self, config, *args, **kwargs):
        super().__init__()
        get = config.get if hasattr(config, "get") else (lambda k, d=None: getattr(config, k, d))
        in_dim = int(get("in_dim"))
        model_data_dir = get("model_data_dir", None) or ""
        weights_file = get("weights_file", "fc7_w.pkl")
        bias_file = get("bias_file", "fc7_b.pkl")
        if not os.path.isabs(weights_file):
            weights_file = os.path.join(model_data_dir, weights_file)
        if not os.path.isabs(bias_file):
            bias_file = os.path.join(model_data_dir, bias_file)
        weights = bias = None
        if os.path.exists(weights_file) and os.path.exists(bias_file):
            with open(weights_file, "rb") as w:
                weights = pickle.load(w)
            with open(bias_file, "rb") as b:
                bias = pickle.load(b)
            out_dim = bias.shape[0]
        else:
            out_dim = int(get("out_dim", in_dim))
            warnings.warn("fc7 weights %s / %s not found (no download in this build): random initialisation" % (weights_file, bias_file))
        self.lc = Linear(in_dim, out_dim)
        if weights is not None:
            self.lc.weight.data.copy_(torch.as_tensor(weights))
            self.lc.bias.data.copy_(torch.as_tensor(bias))
        else:
            self.lc.weight.data.normal_(mean=0.0, std=0.02)
            self.lc.bias.data.zero_()
        self.out_dim = out_dim

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        old_prefix = prefix + "module."            # encoders.py:151-165: checkpoints written through a wrapper module
        for k in list(state_dict.keys()):
            if k.startswith(old_prefix):
                state_dict[k.replace(old_prefix, prefix)] = state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, image):
        return (F32T.relu if F32T.active() else (F32P.relu if F32P.active() else Fn.ReluFn.apply))(self.lc(image))      # encoders.py:177-180
