"""Pins oracle/m4c_oracle.py against the fixture produced by the REAL reference M4C path (`M4C._forward_txt_encoding`,
`_forward_obj_encoding`, `_forward_ocr_encoding`, `_forward_mmt_and_output`, `TextBert`, `MMT`, `OcrPtrNet`,
`PrevPredEmbeddings`, `M4CDecodingBCEWithMaskLoss`, and `M4C.forward`'s greedy decoding loop); see
tests/golden/make_golden.py::make_m4c."""
import numpy as np
import torch

from oracle import m4c_oracle as O
from tests.golden_utils import load_m4c_case


def test_m4c_oracle_matches_reference_forward_loss_and_gradients():
    z, case, cfg, sd, sample = load_m4c_case()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in O.parameter_shapes(cfg).items()}
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.m4c_forward(sd, cfg, dict(sample), training_mode=True, return_all=True)
    for k in ("obj_mmt_in", "ocr_mmt_in", "mmt_seq_output", "scores"):
        np.testing.assert_allclose(out[k].detach().numpy(), z[k], rtol=1e-5, atol=2e-5, err_msg=k)
    loss = O.decoding_bce_with_mask(out["scores"], sample["targets"], sample["train_loss_mask"]).sum()
    assert abs(loss.item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    loss.backward()
    checked = 0
    for gname, norm, gsum in zip(z["grad_names"], z["grad_norms"], z["grad_sums"]):
        key = str(gname)
        g = sd[key].grad
        assert g is not None, key
        if key.endswith(".key.bias") and "ocr_ptr_net" not in key:   # softmax is shift-invariant: noise against noise
            assert norm < 1e-5 and float(g.double().norm()) < 1e-5, key
            continue
        checked += 1
        assert abs(float(g.double().norm()) - norm) <= 1e-4 * norm + 1e-9, key
        assert abs(float(g.double().sum()) - gsum) <= 1e-4 * norm + 1e-7, key
        full = "grad::" + key
        if full in z.files:
            np.testing.assert_allclose(g.numpy(), z[full], rtol=1e-4, atol=1e-6 + 1e-5 * norm, err_msg=key)
    assert checked >= 80
    # [PAD] word row gets no gradient (padding_idx); the classifier's weight gets one from BOTH of its uses
    wg = sd["text_bert.embeddings.word_embeddings.weight"].grad
    assert float(wg[0].abs().max()) == 0.0


def test_m4c_oracle_prefix_lm_mask_structure():
    am = torch.tensor([[1., 1., 0., 1., 0., 0., 0.]])   # 4 encoding positions (one padded), 3 decoding steps
    m = O.prefix_lm_mask(am, 3)[0, 0]
    keep = (m == 0).int()
    assert keep[:, :4].tolist() == [[1, 1, 0, 1]] * 7            # everyone sees the valid encoding positions
    assert keep[:4, 4:].sum() == 0                               # encoding positions never see decoding steps
    assert keep[4:, 4:].tolist() == [[1, 0, 0], [1, 1, 0], [1, 1, 1]]


def test_m4c_oracle_greedy_decoding_matches_reference():
    z, case, cfg, sd, sample = load_m4c_case()
    with torch.no_grad():
        out = O.m4c_forward(sd, cfg, dict(sample), training_mode=False, return_all=True)
    np.testing.assert_array_equal(out["scores"].argmax(-1).numpy(), z["decode_argmax"])
    np.testing.assert_allclose(out["scores"].numpy(), z["decode_scores"], rtol=1e-5, atol=2e-5)


def test_m4c_oracle_greedy_decoding_feeds_back_ocr_copies():
    """Second decoding fixture (sharpened output layers, tests/golden/detweights.py): the reference's greedy sequence mixes
    fixed-vocabulary and OCR-copy indices, so previous predictions >= num_choices go through the OCR half of the gather."""
    from tests.golden import detweights
    z, case, cfg, sd, sample = load_m4c_case()
    am = z["decode2_argmax"]
    V = case["num_choices"]
    assert (am >= V).any() and (am < V).any() and len(np.unique(am)) >= 4      # the fixture really is mixed
    sd2 = {k: torch.from_numpy(v) for k, v in detweights.sharpen_m4c_decoder({k: v.numpy() for k, v in sd.items()}).items()}
    with torch.no_grad():
        out = O.m4c_forward(sd2, cfg, dict(sample), training_mode=False, return_all=True)
    np.testing.assert_array_equal(out["scores"].argmax(-1).numpy(), am)
    np.testing.assert_allclose(out["scores"].numpy(), z["decode2_scores"], rtol=1e-5, atol=1e-4)
