"""Pins oracle/mmft_oracle.py against the fixture produced by the REAL reference MMF Transformer path
(MMFTransformer.forward/preprocess_sample + BaseTransformerBackend.forward + HuggingfaceEmbeddings + BertModelJit
encoder + MLP head + cross_entropy); see tests/golden/make_golden.py::make_mmft."""
import numpy as np
import torch

from oracle import mmft_oracle as O
from tests.golden_utils import load_mmft_case


def test_mmft_oracle_matches_reference_forward_loss_and_gradients():
    z, case, cfg, sd, sample = load_mmft_case()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in O.parameter_shapes(cfg).items()}
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.mmft_forward(sd, cfg, dict(sample), train=False)
    np.testing.assert_allclose(out["scores"].detach().numpy(), z["scores"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(out["sequence_output"].detach().numpy(), z["sequence_output"], rtol=1e-5, atol=5e-6)
    loss = torch.nn.functional.cross_entropy(out["scores"], sample["targets"])
    assert abs(loss.item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    loss.backward()
    alias = O.shared(cfg)
    seen = 0
    for gname, norm, gsum in zip(z["grad_names"], z["grad_norms"], z["grad_sums"]):
        key = alias.get(str(gname), str(gname))
        g = sd[key].grad
        if norm == 0.0:   # the transformer's own position/type tables and pooler are not on MMFT's forward path
            assert g is None or float(g.abs().max()) == 0.0, key
            continue
        seen += 1
        assert g is not None, key
        assert abs(float(g.double().norm()) - norm) <= 1e-4 * norm + 1e-9, key
        assert abs(float(g.double().sum()) - gsum) <= 1e-4 * norm + 1e-7, key
        full = "grad::" + str(gname)
        if full in z.files:
            np.testing.assert_allclose(g.numpy(), z[full], rtol=1e-4, atol=1e-6 + 1e-5 * norm, err_msg=key)
    assert seen >= 40
    # [PAD] rows of the word table receive no gradient (padding_idx) although pad ids occur in the batch
    assert (sample["input_ids"] == 0).any()
    assert float(sd["backend.transformer.embeddings.word_embeddings.weight"].grad[0].abs().max()) == 0.0


def test_reference_state_dict_aliases():
    z, case, cfg, sd, sample = load_mmft_case()
    keys = set(str(k) for k in z["state_dict_keys"])
    for alias, owner in O.shared(cfg).items():
        assert alias in keys and owner in keys


def test_transformer_heads_oracle_matches_reference():
    """`mlm` / `itm` heads (mmf/models/transformers/heads/{mlm,itm}.py): logits, losses, every gradient incl. the tied table's and the
    one handed back to the encoder."""
    import numpy as np
    import torch
    from tests.golden_utils import load_transformer_heads_case
    z, case, sds, inp = load_transformer_heads_case()
    mlm = {k: v.clone().requires_grad_(True) for k, v in sds["mlm"].items()}
    itm = {k: v.clone().requires_grad_(True) for k, v in sds["itm"].items()}
    table = sds["table"]["weight"].clone().requires_grad_(True)
    seq = inp["sequence_output"].clone().requires_grad_(True)
    a = O.mlm_head(mlm, table, seq, inp["labels"])
    b = O.itm_head(itm, seq, inp["is_correct"])
    np.testing.assert_allclose(a["logits"].detach().numpy(), z["mlm_logits"], rtol=1e-5, atol=5e-6)
    assert abs(a["losses"]["masked_lm_loss"].item() - float(z["mlm_loss"])) <= 1e-5 * float(z["mlm_loss"])
    assert abs(b["losses"]["itm_loss"].item() - float(z["itm_loss"])) <= 1e-5 * float(z["itm_loss"])
    (a["losses"]["masked_lm_loss"] + b["losses"]["itm_loss"]).backward()
    np.testing.assert_allclose(seq.grad.numpy(), z["grad_sequence_output"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(table.grad.numpy(), z["grad::table.weight"], rtol=1e-4, atol=1e-7)
    for tag, sd in (("mlm", mlm), ("itm", itm)):
        for k, v in sd.items():
            np.testing.assert_allclose(v.grad.numpy(), z["grad::%s.%s" % (tag, k)], rtol=1e-4, atol=1e-7, err_msg=k)
    blank = torch.full_like(inp["labels"], -1)
    assert O.mlm_head(mlm, table, seq, blank)["losses"]["masked_lm_loss"].item() == 0.0


def test_mrc_head_oracle_matches_reference():
    """`mrc` head (mmf/models/transformers/heads/mrc.py), KL and cross-entropy variants: loss, parameter gradients, encoder gradient."""
    import numpy as np
    from tests.golden_utils import load_transformer_heads_case
    z, case, sds, inp = load_transformer_heads_case()
    for tag, use_kl in (("kl", True), ("ce", False)):
        hsd = {k: v.clone().requires_grad_(True) for k, v in sds["mrc"].items()}
        seq = inp["sequence_output"].clone().requires_grad_(True)
        loss = O.mrc_head(hsd, seq, inp["region_class"], inp["region_mask"], use_kl=use_kl)["losses"]["mrc_loss"]
        assert abs(loss.item() - float(z["mrc_%s_loss" % tag])) <= 1e-5 * float(z["mrc_%s_loss" % tag])
        loss.backward()
        np.testing.assert_allclose(seq.grad.numpy(), z["mrc_%s_grad_sequence_output" % tag], rtol=1e-4, atol=1e-7)
        for k, v in hsd.items():
            np.testing.assert_allclose(v.grad.numpy(), z["grad::mrc_%s.%s" % (tag, k)], rtol=1e-4, atol=1e-7, err_msg=k)


def test_mrfr_and_wra_oracles_match_reference():
    """Oracles for the two remaining UNITER pretraining heads (built on the HIP side in round 3: tests/test_uniter_pretraining_gpu.py; mmf/models/transformers/heads/{mrfr,wra}.py,
    mmf/modules/ot.py), pinned against the reference heads' own run: losses, parameter gradients (incl. the image-embedding weight MRFR
    ties to, transposed) and the gradient handed back to the encoder — through the 50 IPOT iterations for WRA."""
    import numpy as np
    from tests.golden_utils import load_transformer_heads_extra
    z, case, sds, inp = load_transformer_heads_extra()
    hsd = {k: v.clone().requires_grad_(True) for k, v in sds["mrfr"].items()}
    img_w = sds["img"]["weight"].clone().requires_grad_(True)
    seq = inp["sequence_output"].clone().requires_grad_(True)
    loss = O.mrfr_head(hsd, img_w, seq, inp["mrfr_target"], inp["region_mask"])["losses"]["mrfr_loss"]
    assert abs(loss.item() - float(z["mrfr_loss"])) <= 1e-5 * float(z["mrfr_loss"])
    loss.backward()
    np.testing.assert_allclose(seq.grad.numpy(), z["mrfr_grad_sequence_output"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(img_w.grad.numpy(), z["grad::img.weight"], rtol=1e-4, atol=1e-7)
    for k, v in hsd.items():
        np.testing.assert_allclose(v.grad.numpy(), z["grad::mrfr." + k], rtol=1e-4, atol=1e-7, err_msg=k)
    seq = inp["sequence_output"].clone().requires_grad_(True)
    out = O.wra_head(seq, inp["txt_pad"].shape[1], inp["img_pad"].shape[1], inp["txt_pad"], inp["img_pad"], inp["is_correct"])
    assert abs(out["losses"]["wra_loss"].item() - float(z["wra_loss"])) <= 1e-5 * abs(float(z["wra_loss"]))
    out["losses"]["wra_loss"].backward()
    np.testing.assert_allclose(seq.grad.numpy(), z["wra_grad_sequence_output"], rtol=1e-4, atol=1e-7)
