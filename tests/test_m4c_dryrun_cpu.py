"""Host-side dry run of the M4C training step and greedy decoding on CPU: the kernels are replaced by extent / dtype
checkers (tests/native_stub.py), so this exercises the Python half of the path — every autograd Function's forward and
backward, buffer sizes and leading dimensions, the prefix-LM mask hand-off, one gradient per parameter — not the numbers
(those are tests/test_m4c_gpu.py's, against the fixture of the real reference)."""
import torch

from mmf_amd.common.sample import SampleList
from tests import native_stub
from tests.golden_utils import load_m4c_case
from tests.model_utils import build_m4c


def test_m4c_training_step_plumbing():
    z, case, cfg, sd, sample = load_m4c_case()
    model = build_m4c(cfg, sd, device="cpu")
    model.train()
    D = case["D"]
    L = case["T"] + case["O"] + case["N"] + D
    with native_stub.installed() as calls:
        out = model(SampleList(sample))
        assert tuple(out["scores"].shape) == (case["B"], D, case["num_choices"] + case["N"]) and out["scores"].dtype == torch.float32
        (key, loss), = out["losses"].items()
        assert key == "train/textvqa/m4c_decoding_bce_with_mask" and tuple(loss.shape) == (1,)
        loss.sum().backward()
    fwd = [c for c in calls if c[0] == "attention_fwd"]
    bwd = [c for c in calls if c[0] == "attention_bwd"]
    n_txt, n_mmt = cfg["text_num_hidden_layers"], cfg["num_hidden_layers"]
    assert len(fwd) == len(bwd) == n_txt + n_mmt
    assert [c[-1] for c in fwd] == [0] * n_txt + [D] * n_mmt                     # causal tail only inside the MMT
    assert all(c[3] == L for c in fwd[n_txt:]) and all(c[3] == case["T"] for c in fwd[:n_txt])
    assert sorted(c[-1] for c in bwd) == [0] * n_txt + [D] * n_mmt
    missing = [n for n, p in model.named_parameters() if p.grad is None]
    assert not missing, missing
    for n, p in model.named_parameters():
        assert p.grad.shape == p.shape and p.grad.dtype == torch.float32, n


def test_m4c_greedy_decoding_plumbing():
    z, case, cfg, sd, sample = load_m4c_case()
    model = build_m4c(cfg, sd, device="cpu")
    model.eval()
    D = case["D"]
    with native_stub.installed() as calls, torch.no_grad():
        out = model(SampleList(sample))
    assert tuple(out["scores"].shape) == (case["B"], D, case["num_choices"] + case["N"])
    fwd = [c for c in calls if c[0] == "attention_fwd"]
    n_txt, n_mmt = cfg["text_num_hidden_layers"], cfg["num_hidden_layers"]
    E = case["T"] + case["O"] + case["N"]
    # incremental decoding: text_bert once, the encoder positions ONCE through the multimodal transformer (E queries x E keys),
    # then one new row per sample and step against the K|V cache (1 query, E + i + 1 keys) — the reference re-encodes all
    # E + D positions D times (m4c.py:297-305)
    assert len(fwd) == n_txt + n_mmt + D * n_mmt
    assert all(c[3] == E and c[4] == E for c in fwd[n_txt:n_txt + n_mmt])
    steps = fwd[n_txt + n_mmt:]
    assert [(c[3], c[4]) for c in steps] == [(1, E + i + 1) for i in range(D) for _ in range(n_mmt)]
    assert all(c[-1] == 0 for c in fwd)               # no causal tail needed: a step only addresses the keys it may see
    # the reference-style loop stays available for A/B checks
    from tests.model_utils import m4c_model_config
    model2 = build_m4c(cfg, sd, device="cpu", kv_cached_decode=False)
    model2.eval()
    with native_stub.installed() as calls2, torch.no_grad():
        model2(SampleList(sample))
    assert len([c for c in calls2 if c[0] == "attention_fwd"]) == n_txt + D * n_mmt


import pytest


@pytest.mark.parametrize("switch", ["remove_ocr_fasttext", "remove_ocr_phoc", "remove_ocr_frcn", "remove_ocr_semantics", "remove_ocr_bbox"])
def test_m4c_ocr_ablation_switches_plumbing(switch):
    """The `remove_ocr_*` switches of m4c.py:121-125,229-241: the step still runs end to end and the parameters upstream of a
    removed input get no (or an all-zero-input) gradient path without breaking the others."""
    z, case, cfg, sd, sample = load_m4c_case()
    from tests.model_utils import m4c_model_config
    base = m4c_model_config(cfg)
    ocr = dict(base["ocr"]); ocr[switch] = True
    model = build_m4c(cfg, sd, device="cpu", ocr=ocr)
    assert getattr(model, switch) is True
    model.train()
    with native_stub.installed():
        out = model(SampleList(sample))
        (key, loss), = out["losses"].items()
        loss.sum().backward()
    for n, p in model.named_parameters():
        if n.startswith("ocr_faster_rcnn_fc7.") and switch in ("remove_ocr_frcn", "remove_ocr_semantics"):
            continue                                    # the appearance feature is cut off: no gradient reaches fc7
        assert p.grad is not None, n


def test_m4c_identity_text_projection_plumbing():
    """text_bert as wide as the MMT (the real configuration): `text_bert_out_linear` is nn.Identity (m4c.py:88-98)."""
    z, case, cfg, sd, sample = load_m4c_case()
    cfg = dict(cfg, text_hidden_size=cfg["hidden_size"], text_num_attention_heads=cfg["num_attention_heads"],
               text_intermediate_size=cfg["intermediate_size"])
    model = build_m4c(cfg, None, device="cpu")
    assert isinstance(model.text_bert_out_linear, torch.nn.Identity)
    model.train()
    with native_stub.installed():
        out = model(SampleList(sample))
        (key, loss), = out["losses"].items()
        loss.sum().backward()
    assert all(p.grad is not None for p in model.parameters())
