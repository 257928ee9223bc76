"""M4C behind the reference's plug-in boundary, on CPU (no kernel launches): registry entry, construction through
`build_model` with the registry keys the reference reads, parameter tree == the reference's state dict (as recorded in
the fixture), optimizer groups of `get_optimizer_parameters` (m4c.py:307-329), the loss registration, and the prefix-LM
mask hand-off to the attention op."""
import pytest
import torch

import mmf_amd  # noqa: F401
from mmf_amd import functional as Fn
from mmf_amd.common.registry import registry
from mmf_amd.modules.hf_layers import additive_key_mask
from mmf_amd.utils.configuration import Config
from oracle import m4c_oracle as O
from tests.golden_utils import load_m4c_case
from tests.model_utils import build_m4c


def test_m4c_is_registered_and_needs_the_dataset_registry_entries():
    cls = registry.get_model_class("m4c")
    assert cls is not None and cls.config_path() == "configs/models/m4c/defaults.yaml"
    assert registry.get_loss_class("m4c_decoding_bce_with_mask") is not None
    assert registry.get_encoder_class("finetune_faster_rcnn_fpn_fc7") is not None


def test_m4c_parameter_tree_matches_the_reference_state_dict():
    z, case, cfg, sd, sample = load_m4c_case()
    model = build_m4c(cfg, sd, device="cpu")
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    ref = {k: tuple(v) for k, v in O.parameter_shapes(cfg).items()}
    assert mine == ref
    assert set(mine) == {str(n) for n in z["param_names"]}
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_m4c_default_config_builds_the_textvqa_shape():
    """configs/models/m4c/defaults.yaml: 3-layer text BERT, 4-layer MMT, 768 wide, 3002-wide OCR feature, Identity projection."""
    registry.register("config", Config({"datasets": "textvqa"}))
    registry.register("textvqa_num_final_outputs", 5050)
    registry.register("textvqa_answer_processor", Config({"BOS_IDX": 1}))
    with pytest.warns(UserWarning):
        model = registry.get_model_class("m4c")(Config({"model": "m4c", "text_bert_init_from_bert_base": False}))
        model.build()
    assert isinstance(model.text_bert_out_linear, torch.nn.Identity)
    assert len(model.text_bert.encoder.layer) == 3 and len(model.mmt.encoder.layer) == 4
    assert tuple(model.linear_ocr_feat_to_mmt_in.weight.shape) == (768, 3002)
    assert tuple(model.classifier.module.weight.shape) == (5000, 768)
    assert tuple(model.obj_faster_rcnn_fc7.lc.weight.shape) == (2048, 2048)
    n = sum(p.numel() for p in model.parameters())
    assert 90_000_000 < n < 100_000_000, n


def test_m4c_optimizer_groups_follow_the_reference():
    z, case, cfg, sd, sample = load_m4c_case()
    model = build_m4c(cfg, None, device="cpu", text_bert_init_from_bert_base=False)
    groups = model.get_optimizer_parameters(Config({"optimizer": {"params": {"lr": 1e-4}}}))
    # default-lr group first, then obj fc7, ocr fc7 (lr_scale_frcn) and the MMT (lr_scale_mmt); text_bert joins only when
    # it starts from BERT-base (m4c.py:75-82)
    assert "lr" not in groups[0]
    assert [round(g["lr"] / 1e-4, 6) for g in groups[1:]] == [0.1, 0.1, 1.0]
    seen = [id(p) for g in groups for p in g["params"]]
    assert len(seen) == len(set(seen)) == len(list(model.parameters()))
    default_names = {n for n, p in model.named_parameters() if any(p is q for q in groups[0]["params"])}
    assert any(n.startswith("text_bert.") for n in default_names) and any(n.startswith("classifier.") for n in default_names)
    assert not any(n.startswith("mmt.") or "faster_rcnn_fc7" in n for n in default_names)


def test_prefix_lm_mask_is_handed_to_the_attention_op_unmaterialised():
    key = torch.zeros(2, 10)
    m = additive_key_mask(Fn.PrefixLMMask(key.view(2, 1, 1, 10), 3), 2, 10)
    assert isinstance(m, Fn.PrefixLMMask) and m.causal_tail == 3 and tuple(m.key_mask.shape) == (2, 10)
    assert Fn._split_mask(m)[1] == 3 and Fn._split_mask(key) == (key, 0)
    # a materialised [B, 1, S, S] mask (what MMT.forward builds, m4c.py:424-440) is accepted as it is: the kernels read it per (query, key) pair
    m3 = additive_key_mask(torch.zeros(2, 1, 10, 10), 2, 10)
    assert tuple(m3.shape) == (2, 10, 10) and m3.dtype == torch.float32 and m3.is_contiguous()
    m4 = additive_key_mask(torch.zeros(2, 4, 10, 10), 2, 10)      # one mask per head (mmf_attn_desc.mask_head_stride, round 6): handed on as it is
    assert tuple(m4.shape) == (2, 4, 10, 10) and m4.dtype == torch.float32 and m4.is_contiguous()
    with pytest.raises(ValueError):
        additive_key_mask(torch.zeros(2, 4, 10, 11), 2, 10)
    with pytest.raises(ValueError):
        additive_key_mask(torch.zeros(2, 1, 9, 10), 2, 10)
