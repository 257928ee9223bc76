"""`bypass_transformer: true` of VisualBERT (mmf/models/visual_bert.py:52-56, 116-141) on the GPU against the fixture recorded from the
reference's own run (tests/golden/make_visual_bert_bypass.py).

First run on hardware: round 3, green; part of the default `-m gpu` run since.  The path is host-side glue over kernels that are verified
on their own (encoder layers, row concat, pooler): its oracle is pinned (tests/test_oracle_golden.py), its host logic dry-runs with the
reference's gradient pattern (tests/test_dryrun_models_cpu.py[visual_bert_bypass])."""

import numpy as np
import pytest
import torch

from mmf_amd.common.sample import SampleList
from tests.golden_utils import load_bypass_case
from tests.model_utils import build_visual_bert, sample_to

pytestmark = pytest.mark.gpu
TOL = 5e-2


def test_bypass_transformer_golden_forward_loss_and_gradients():
    z, case, cfg, sd, sample = load_bypass_case()
    model = build_visual_bert(cfg, sd, bypass_transformer=True, pooler_strategy="default")
    model.eval()
    out = model(SampleList(sample_to(sample, "cuda")))
    np.testing.assert_allclose(out["scores"].detach().float().cpu().numpy(), z["scores"], rtol=TOL, atol=TOL)
    (key, loss), = out["losses"].items()
    assert abs(loss.item() - float(z["loss"])) <= TOL * abs(float(z["loss"]))
    loss.sum().backward()
    params = dict(model.named_parameters())
    bad = {}
    for gname, norm in zip(z["grad_names"], z["grad_norms"]):
        gname = str(gname)
        p = params[gname]
        if norm == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, gname
            continue
        if gname.endswith("self.key.bias"):
            continue
        assert p.grad is not None, gname
        e = abs(float(p.grad.double().norm()) - norm) / norm
        if e > TOL:
            bad[gname] = e
    assert not bad, bad
