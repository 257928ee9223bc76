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


import pytest


@pytest.mark.parametrize("task", ["mlm", "itm", "mrc"])
def test_uniter_pretraining_oracle_matches_reference(task):
    """UNITERForPretraining (mmf/models/uniter.py:350-618) for the tasks whose heads are built: what the reference's preprocessing hands
    the encoder and the head (bit for bit, given the region mask it drew), the loss, and every gradient."""
    from tests.golden_utils import load_uniter_pretraining_case
    z, case, cfg, sd, sample = load_uniter_pretraining_case()
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    region_masks = torch.from_numpy(z["mrc_pre_image_mask"]) if task == "mrc" else None
    out = O.uniter_pretraining_forward(sd, cfg, sample, task, region_masks)
    pre = out["preprocessed"]
    assert torch.equal(pre["input_ids"], torch.from_numpy(z[task + "_pre_input_ids"]))
    assert torch.equal(pre["image_feat"], torch.from_numpy(z[task + "_pre_image_feat"]))
    assert torch.equal(pre["image_mask"].long(), torch.from_numpy(z[task + "_pre_image_mask"]))
    if task == "mrc":
        assert torch.equal(pre["region_class"], torch.from_numpy(z["mrc_pre_region_class"]))
        assert torch.equal(pre["image_region_mask"].long(), torch.from_numpy(z["mrc_pre_image_region_mask"]))
    if task == "mlm":
        assert torch.equal(pre["mlm_labels"]["combined_labels"], torch.from_numpy(z["mlm_pre_combined_labels"]))
    (key, loss), = out["losses"].items()
    assert key == str(z[task + "_loss_key"]) and abs(loss.item() - float(z[task + "_loss"])) <= 1e-5 * float(z[task + "_loss"])
    loss.backward()
    for gname, norm in zip(z[task + "_grad_names"], z[task + "_grad_norms"]):
        key = "uniter." + str(gname)
        if key.endswith("predictions.decoder.bias"):
            continue
        g = sd[key].grad
        if norm == 0.0:
            assert g is None or float(g.abs().max()) == 0.0, key
            continue
        if key.endswith("self.key.bias"):
            continue
        assert g is not None and abs(float(g.double().norm()) - norm) <= 1e-4 * norm + 1e-9, key
        full = "grad::%s::%s" % (task, str(gname))
        if full in z.files:
            np.testing.assert_allclose(g.numpy(), z[full], rtol=1e-4, atol=1e-6 + 1e-5 * norm, err_msg=key)


@pytest.mark.parametrize("task", ["mrfr", "wra"])
def test_uniter_default_task_list_oracle_matches_reference(task):
    """The wrapper built with the reference's DEFAULT tasks (mlm, itm, mrc, mrfr, wra; uniter.py:36-39): `_preprocess_mrfr` / `_preprocess_wra`,
    the MRFR head tied to the image embedding's weight and the WRA head (50 IPOT steps), against the reference's own run."""
    from tests.golden_utils import load_uniter_pretraining_all_case
    z, case, cfg, sd, sample = load_uniter_pretraining_all_case()
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not k.endswith("linear_proj_weight")}     # (one tensor, listed twice)
    region_masks = torch.from_numpy(z["mrfr_pre_image_mask"]) if task == "mrfr" else None
    out = O.uniter_pretraining_forward(sd, cfg, sample, task, region_masks)
    pre = out["preprocessed"]
    assert torch.equal(pre["input_ids"], torch.from_numpy(z[task + "_pre_input_ids"]))
    assert torch.equal(pre["image_feat"], torch.from_numpy(z[task + "_pre_image_feat"]))
    if task == "mrfr":
        assert torch.equal(pre["mrfr_region_target"], torch.from_numpy(z["mrfr_pre_region_target"]))
        assert torch.equal(pre["mrfr_region_mask"].long(), torch.from_numpy(z["mrfr_pre_region_mask"]))
    else:
        assert torch.equal(pre["wra_info"]["txt_pad"].long(), torch.from_numpy(z["wra_pre_txt_pad"]))
        assert torch.equal(pre["wra_info"]["img_pad"].long(), torch.from_numpy(z["wra_pre_img_pad"]))
    (key, loss), = out["losses"].items()
    assert key == str(z[task + "_loss_key"]) and abs(loss.item() - float(z[task + "_loss"])) <= 1e-5 * abs(float(z[task + "_loss"]))
    loss.backward()
    checked = 0
    for gname, norm in zip(z[task + "_grad_names"], z[task + "_grad_norms"]):
        key = "uniter." + str(gname)
        if key.endswith("predictions.decoder.bias") or key.endswith("linear_proj_weight"):      # (the tied weight is listed under img_linear)
            continue
        g = sd[key].grad
        if norm == 0.0:
            assert g is None or float(g.abs().max()) == 0.0, key
            continue
        if key.endswith("self.key.bias"):
            continue
        assert g is not None, key
        assert abs(float(g.double().norm()) - norm) <= 1e-4 * norm + 1e-9, (key, float(g.double().norm()), norm)
        checked += 1
    assert checked >= 30
