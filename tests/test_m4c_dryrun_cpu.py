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
    # text_bert once (deterministic in eval mode), the multimodal transformer once per decoding step (m4c.py:297-305)
    assert len(fwd) == cfg["text_num_hidden_layers"] + D * cfg["num_hidden_layers"]
