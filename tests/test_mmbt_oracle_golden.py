"""Pins oracle/mmbt_oracle.py against the fixture produced by the REAL reference MMBT path
(MMBTBase.forward + MMBTModel + ModalEmbeddings + BertModelJit + MMBTForClassification.forward + cross_entropy)."""
import numpy as np
import pytest
import torch

from oracle import mmbt_oracle as O
from tests.golden_utils import load_mmbt_case


# mmbt_decoder64: `is_decoder: true` (mmbt.py:244-272 — padding mask times a causal mask over modal + text positions; the layers then also own an
# unused crossattention block, hf_layers.py:268-292: parameters without a gradient)
@pytest.mark.parametrize("name", ["mmbt_small64", "mmbt_decoder64"])
def test_mmbt_oracle_matches_reference_forward_loss_and_gradients(name):
    z, case, cfg, sd, sample = load_mmbt_case(name)
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in O.parameter_shapes(cfg).items()}
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.mmbt_forward(sd, cfg, dict(sample), train=False)
    np.testing.assert_allclose(out["scores"].detach().numpy(), z["scores"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(out["sequence_output"].detach().numpy(), z["sequence_output"], rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(out["pooled_output"].detach().numpy(), z["pooled_output"], rtol=1e-5, atol=5e-6)
    loss = O.cross_entropy(out["scores"], sample["targets"])
    assert abs(loss.item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    loss.backward()
    for gname, norm, gsum in zip(z["grad_names"], z["grad_norms"], z["grad_sums"]):
        key = str(gname)[len("model."):]
        key = O.SHARED.get(key, key)
        g = sd[key].grad
        if norm == 0.0 and "crossattention" in key:      # never called by BertLayerJit.forward: no gradient on either side
            assert g is None or float(g.abs().max()) == 0.0, key
            continue
        assert g is not None, key
        assert abs(float(g.double().norm()) - norm) <= 1e-4 * norm + 1e-9, key
        assert abs(float(g.double().sum()) - gsum) <= 1e-4 * norm + 1e-7, key
        full = "grad::" + str(gname)
        if full in z.files:
            np.testing.assert_allclose(g.numpy(), z[full], rtol=1e-4, atol=1e-6 + 1e-5 * norm, err_msg=key)


def test_reference_state_dict_lists_the_shared_tables_twice():
    z, case, cfg, sd, sample = load_mmbt_case()
    keys = set(str(k)[len("model."):] for k in z["state_dict_keys"])
    for alias, owner in O.SHARED.items():
        assert alias in keys and owner in keys


def test_mmbt_pretraining_oracle_matches_reference():
    """MMBTForPreTraining.forward (mmbt.py:479-523), masked-LM branch: logits over all positions, loss over the text positions,
    gradients (the tied word-embedding table collects the gather and the decoder gradient)."""
    from tests.golden_utils import load_mmbt_pretraining_case
    z, case, cfg, sd, sample = load_mmbt_pretraining_case()
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.mmbt_pretraining_forward(sd, cfg, dict(sample))
    np.testing.assert_allclose(out["logits"].detach().numpy(), z["logits"], rtol=1e-5, atol=5e-6)
    (key, loss), = out["losses"].items()
    assert key == str(z["loss_key"]) == "hateful_memes/train/masked_lm_loss"
    assert abs(loss.item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    loss.backward()
    for gname, norm in zip(z["grad_names"], z["grad_norms"]):
        key = str(gname)[len("model."):]
        key = O.SHARED.get(key, key)
        g = sd[key].grad
        if norm == 0.0:        # pooler, next-sentence head: outside the loss
            assert g is None or float(g.abs().max()) == 0.0, key
            continue
        if key.endswith("self.key.bias"):
            continue
        assert g is not None and abs(float(g.double().norm()) - norm) <= 1e-4 * norm + 1e-9, key
        full = "grad::" + str(gname)
        if full in z.files:
            np.testing.assert_allclose(g.numpy(), z[full], rtol=1e-4, atol=1e-6 + 1e-5 * norm, err_msg=key)
