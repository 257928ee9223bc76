"""Pins oracle/uniter_oracle.py against the fixture produced by the REAL reference UNITER path (UNITER.forward /
add_custom_params / add_pos_feat, UNITERForClassification -> _infer_with_heads, UNITERModelBase, UNITERImageEmbeddings, MLP
head, logit_bce); see tests/golden/make_golden.py::make_uniter."""
import numpy as np
import torch

from oracle import uniter_oracle as O
from oracle.visual_bert_oracle import logit_bce
from tests.golden_utils import load_uniter_case


def test_uniter_oracle_matches_reference_forward_loss_and_gradients():
    z, case, cfg, sd, sample = load_uniter_case()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in O.parameter_shapes(cfg).items()}
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.uniter_forward(sd, cfg, dict(sample), train=False)
    np.testing.assert_allclose(out["img_pos_feat"].numpy(), z["img_pos_feat"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(out["scores"].detach().numpy(), z["scores"], rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(out["sequence_output"].detach().numpy(), z["sequence_output"], rtol=1e-5, atol=1e-5)
    loss = logit_bce(out["scores"], sample["targets"])
    assert str(z["loss_key"]) == "train/vqa2/logit_bce"
    assert abs(loss.item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    loss.backward()
    checked = 0
    for gname, norm, gsum in zip(z["grad_names"], z["grad_norms"], z["grad_sums"]):
        key = str(gname)
        g = sd[key].grad
        if norm == 0.0:   # BertModel's own pooler is kept but never called (uniter.py:150)
            assert key.startswith("uniter.uniter.pooler.") and (g is None or float(g.abs().max()) == 0.0), key
            continue
        assert g is not None, key
        if key.endswith(".key.bias"):
            assert norm < 1e-6 and float(g.double().norm()) < 1e-6, key
            continue
        checked += 1
        assert abs(float(g.double().norm()) - norm) <= 1e-4 * norm + 1e-9, key
        assert abs(float(g.double().sum()) - gsum) <= 1e-4 * norm + 1e-7, key
        full = "grad::" + key
        if full in z.files:
            np.testing.assert_allclose(g.numpy(), z[full], rtol=1e-4, atol=1e-6 + 1e-5 * norm, err_msg=key)
    assert checked >= 40
    # the mask embedding's row 0 gets no gradient (padding_idx = 0), row 1 does (added to every valid region)
    mg = sd["uniter.uniter.img_embeddings.mask_embedding.weight"].grad
    assert float(mg[0].abs().max()) == 0.0 and float(mg[1].abs().max()) > 0.0
