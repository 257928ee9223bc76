"""Pins oracle/vilbert_oracle.py against the fixture produced by the REAL reference ViLBERT path (ViLBERT.forward ->
ViLBERTForClassification.forward -> ViLBERTBase -> BertEncoder / BertConnectionLayer / BertBiAttention ... + logit_bce);
see tests/golden/make_golden.py::make_vilbert."""
import numpy as np
import pytest
import torch

from oracle import vilbert_oracle as O
from oracle.visual_bert_oracle import logit_bce
from tests.golden_utils import load_vilbert_case


# vilbert_dyn: dynamic_attention gates (vilbert.py:199-212); vilbert_fixed: fixed_t_layer 2 / fixed_v_layer 1 (:625-666 — one detached layer, one skipped);
# vilbert_pairs: in_batch_pairs (:678-710 — every text against every image, B^2 score rows); vilbert_fast: fast_mode (:712-723 — one text, B images)
@pytest.mark.parametrize("name", ["vilbert_small", "vilbert_dyn", "vilbert_fixed", "vilbert_pairs", "vilbert_fast"])
def test_vilbert_oracle_matches_reference_forward_loss_and_gradients(name):
    z, case, cfg, sd, sample = load_vilbert_case(name)
    assert any("dyLinear_q" in k for k in sd) == (name == "vilbert_dyn")
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in O.parameter_shapes(cfg).items()}
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.vilbert_forward(sd, cfg, dict(sample), train=False)
    np.testing.assert_allclose(out["scores"].detach().numpy(), z["scores"], rtol=1e-5, atol=5e-6)
    for k in ("sequence_output_t", "sequence_output_v", "pooled_output_t", "pooled_output_v"):
        np.testing.assert_allclose(out[k].detach().numpy(), z[k], rtol=1e-5, atol=1e-5, err_msg=k)
    loss = logit_bce(out["scores"], sample["targets"])
    assert abs(loss.item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    loss.backward()
    checked = 0
    for gname, norm, gsum in zip(z["grad_names"], z["grad_norms"], z["grad_sums"]):
        key = str(gname)[len("model."):]
        g = sd[key].grad
        if norm == 0.0:   # biOutput.q_dense1/2 are declared but never used (vilbert.py:486,493); vilbert_fixed: the detached / skipped layers
            frozen = name == "vilbert_fixed" and (".encoder.layer.0." in key or ".encoder.layer.1." in key or ".encoder.v_layer.0." in key
                                                  or key.startswith("bert.embeddings.") or key.startswith("bert.v_embeddings."))
            assert ("q_dense" in key or frozen) and (g is None or float(g.abs().max()) == 0.0), key
            continue
        checked += 1
        assert g is not None, key
        if key.endswith(".key.bias") or key.endswith("key1.bias") or key.endswith("key2.bias"):
            # identically zero in exact arithmetic (a per-query constant shift cancels in the softmax): fp32 noise on both sides
            assert norm < 1e-6 and float(g.double().norm()) < 1e-6, key
            continue
        assert abs(float(g.double().norm()) - norm) <= 1e-4 * norm + 1e-9, key
        assert abs(float(g.double().sum()) - gsum) <= 1e-4 * norm + 1e-7, key
        full = "grad::" + str(gname)
        if full in z.files:
            np.testing.assert_allclose(g.numpy(), z[full], rtol=1e-4, atol=1e-6 + 1e-5 * norm, err_msg=key)
    # 8 unused q_dense tensors; vilbert_fixed: + the embeddings (5 + 6), text layers 0 (detached) and 1 (skipped) and visual layer 0: 3 x 16
    assert checked == len(sd) - 8 - (5 + 6 + 3 * 16 if name == "vilbert_fixed" else 0)


def test_vilbert_oracle_nlvr2_matches_reference():
    """Two images per sample (vilbert.py:1369-1394) and the paired head (:1262-1265, 1322-1323)."""
    import torch.nn.functional as F
    z, case, cfg, sd, sample = load_vilbert_case("vilbert_nlvr2")
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in O.parameter_shapes(cfg).items()}
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.vilbert_forward(sd, cfg, dict(sample), train=False)
    np.testing.assert_allclose(out["scores"].detach().numpy(), z["scores"], rtol=1e-5, atol=5e-6)
    loss = F.cross_entropy(out["scores"], sample["targets"])
    assert abs(loss.item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    loss.backward()
    for gname, norm in zip(z["grad_names"], z["grad_norms"]):
        key = str(gname)[len("model."):]
        g = sd[key].grad
        if norm == 0.0 or key.endswith(".key.bias") or key.endswith("key1.bias") or key.endswith("key2.bias"):
            continue
        assert g is not None and abs(float(g.double().norm()) - norm) <= 1e-4 * norm + 1e-9, key




@pytest.mark.parametrize("visual_target", [0, 1, 2])        # 2: NCE with the recorded negatives (vilbert.py:1158-1227)
def test_vilbert_pretraining_oracle_matches_reference(visual_target):
    """ViLBERTForPretraining (vilbert.py:1054-1240; visual_target 0: masked-region KL, 1: masked-region regression, round 3): both losses and
    every gradient against the reference's own run."""
    from tests.golden_utils import load_vilbert_pretraining_case
    z, case, cfg, sd, sample = load_vilbert_pretraining_case(visual_target)
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in O.parameter_shapes(cfg).items()}
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.vilbert_pretraining_forward(sd, cfg, dict(sample))
    ref = dict(zip((str(k) for k in z["loss_keys"]), z["loss_values"]))
    assert set(out["losses"]) == set(ref) == {"coco/train/masked_lm_loss", "coco/train/masked_img_loss"}
    for k, v in out["losses"].items():
        assert tuple(v.shape) == (1,) and abs(v.item() - ref[k]) <= 1e-5 * abs(ref[k]), k
    sum(v.sum() for v in out["losses"].values()).backward()
    for gname, norm in zip(z["grad_names"], z["grad_norms"]):
        key = str(gname)[len("model."):]
        g = sd[key].grad
        if norm == 0.0:       # poolers, bi_seq_relationship, biOutput.q_dense*: outside both losses
            assert g is None or float(g.abs().max()) == 0.0, key
            continue
        if key.endswith(".key.bias") or key.endswith("key1.bias") or key.endswith("key2.bias"):
            continue
        assert g is not None and abs(float(g.double().norm()) - norm) <= 1e-4 * norm + 1e-9, key
        full = "grad::" + str(gname)
        if full in z.files:
            np.testing.assert_allclose(g.numpy(), z[full], rtol=1e-4, atol=1e-6 + 1e-5 * norm, err_msg=key)
