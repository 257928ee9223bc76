"""The reference's own encoder tests (tests/modules/test_encoders.py:17-71, the encoders that are on the fusion path) and MMBT's
modal-end-token test (tests/models/test_mmbt.py:63-98), ported: construction through `from_params`, the registry / factory
resolution of `{type, params}` configs, and what `extract_modal_end_token` does to the text.  Forward passes run in
tests/test_encoders_gpu.py (there is no CPU path)."""
import tempfile
import warnings

import pytest
import torch
from torch import nn

import mmf_amd  # noqa: F401
from mmf_amd.common.registry import registry
from mmf_amd.modules import encoders
from mmf_amd.utils.configuration import Config


def _test_init(cls, **params):
    encoder = cls.from_params(**params)
    assert isinstance(encoder, nn.Module)
    return encoder


def test_finetune_faster_rcnn_fpn_fc7():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # (no download here: random init + a warning instead of the detectron pickles)
        enc = _test_init(encoders.FinetuneFasterRcnnFpnFc7, in_dim=2048, model_data_dir=tempfile.TemporaryDirectory().name)
    assert enc.lc.weight.shape == (2048, 2048) and enc.out_dim == 2048


def test_transformer_encoder():
    enc = _test_init(encoders.TransformerEncoder)
    assert enc.embeddings.word_embeddings.weight.size(1) == 768 and enc.embeddings.word_embeddings.weight.size(0) == 30522
    assert enc.config.num_hidden_layers == 12 and type(enc.module).__name__ == "BertModelJit"


def test_multimodal_encoder_base():
    _test_init(encoders.MultiModalEncoderBase)


def test_identity():
    enc = _test_init(encoders.IdentityEncoder, in_dim=256)
    assert enc.in_dim == enc.out_dim == 256
    x = torch.rand(3, 256)
    assert enc(x) is x


def test_registry_and_factories_resolve_the_reference_config_forms():
    assert registry.get_encoder_class("transformer") is encoders.TransformerEncoder
    assert registry.get_encoder_class("identity") is encoders.IdentityEncoder
    assert registry.get_encoder_class("finetune_faster_rcnn_fpn_fc7") is encoders.FinetuneFasterRcnnFpnFc7
    small = dict(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, vocab_size=50)
    e = encoders.build_encoder(Config(type="transformer", params=Config(small)))                # mmf/utils/build.py:524-533
    assert isinstance(e, encoders.TransformerEncoder) and e.config.hidden_size == 128
    e = encoders.build_encoder(Config(name="identity", in_dim=9))                               # structured form, :535-537
    assert e.out_dim == 9
    m = encoders.build_text_encoder(Config(type="transformer", params=Config(small, num_segments=4)))     # build.py:495-503 -> .module
    assert type(m).__name__ == "BertModelJit" and m.embeddings.token_type_embeddings.weight.shape == (4, 128)
    f = encoders.build_image_encoder(Config(type="identity", params=Config(in_dim=72)), direct_features=True)
    assert f.in_dim == f.out_dim == 72
    with pytest.raises(NotImplementedError, match="CNN feature extractor"):
        encoders.build_image_encoder(Config(type="resnet152", params=Config()), direct_features=False)
    with pytest.raises(NotImplementedError):
        encoders.build_text_encoder(Config(type="embedding", params=Config()))


def test_modal_end_token():
    """tests/models/test_mmbt.py:63-98: the last real token (<sep>) becomes the modal end token, the text shifts left by one."""
    from tests.golden_utils import load_mmbt_case
    from tests.model_utils import mmbt_model_config
    from mmf_amd.common.sample import SampleList
    from mmf_amd.utils.build import build_model
    z, case, cfg, sd, sample = load_mmbt_case()
    model = build_model(mmbt_model_config(cfg))
    CLS, PAD, SEP, size = 0, 1, 2, 128
    g = torch.Generator().manual_seed(3)
    input_ids = torch.randint(low=3, high=200, size=(size,), generator=g).long()
    input_mask = torch.ones(size).long()
    input_ids[0] = CLS
    length = int(torch.randint(low=2, high=size - 1, size=(1,), generator=g))
    input_ids[length] = SEP
    input_ids[length + 1:] = PAD
    input_mask[length + 1:] = 0
    sl = SampleList(dict(input_ids=input_ids.clone().unsqueeze(0), input_mask=input_mask.clone().unsqueeze(0),
                         segment_ids=torch.zeros(1, size).long()))
    with torch.no_grad():
        actual = model.model.bert.extract_modal_end_token(sl)
    assert torch.equal(actual, torch.zeros([1]).fill_(SEP).long())
    assert torch.equal(sl["input_ids"][0, :-1], input_ids[1:])
    assert int(sl["input_mask"].sum()) == int(input_mask.sum()) - 1
