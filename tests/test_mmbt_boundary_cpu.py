"""MMBT host mirror, CPU side: the registered model builds through MMF's build path with the reference's
parameter tree (state-dict keys compared against the ones recorded from the real reference, incl. the modal
encoder's aliases of the text embedding tables), and everything off the built path raises instead of
silently degrading."""
import pytest
import torch

from oracle import mmbt_oracle as O
from tests.golden_utils import load_mmbt_case
from tests.model_utils import build_mmbt, mmbt_model_config
from mmf_amd.common.registry import registry
from mmf_amd.utils.build import build_model


@pytest.mark.parametrize("name", ["mmbt_small64", "mmbt_decoder64"])      # decoder mode: the layers also own the reference's crossattention blocks
def test_registered_and_state_dict_matches_reference_tree(name):
    z, case, cfg, sd, sample = load_mmbt_case(name)
    assert registry.get_model_class("mmbt") is not None
    assert registry.get_loss_class("cross_entropy") is not None
    model = build_mmbt(cfg, sd, O.SHARED, device="cpu")
    ours = set(model.state_dict().keys())
    ref = {str(k) for k in z["state_dict_keys"] if not (str(k).endswith("position_ids") or str(k).endswith("embeddings.token_type_ids"))}
    assert ours == ref, (sorted(ours - ref)[:5], sorted(ref - ours)[:5])
    m = model.model.bert.mmbt
    assert m.modal_encoder.word_embeddings.weight is m.transformer.embeddings.word_embeddings.weight
    assert m.modal_encoder.LayerNorm.weight is m.transformer.embeddings.LayerNorm.weight
    # shared parameters are listed once for the optimizer
    names = [n for n, _ in model.named_parameters()]
    assert len(names) == len(set(names)) == len(O.parameter_shapes(cfg))


def test_unbuilt_variants_raise_and_built_encoders_resolve():
    """Raw-image CNN encoders are out of scope (they raise); the encoders of mmf/modules/encoders.py that ARE on the path resolve the
    reference's way: `text_encoder: {type: transformer}` -> TransformerEncoder(...).module (BertModelJit), `num_segments` re-sizes
    the token-type table (encoders.py:567-578), `modal_encoder: {type: finetune_faster_rcnn_fpn_fc7}` of the hateful-memes
    with_features config -> the trainable fc7 layer ahead of the projection."""
    z, case, cfg, sd, sample = load_mmbt_case()
    with pytest.raises(NotImplementedError, match="CNN feature extractor"):
        build_model(mmbt_model_config(cfg, direct_features_input=False, modal_encoder=dict(type="resnet152", params={})))
    model = build_model(mmbt_model_config(cfg, training_head_type="pretraining", losses=[]))       # MMBTForPreTraining is built
    assert type(model.model).__name__ == "MMBTForPreTraining"
    assert model.model.cls.predictions.decoder.weight is model.model.bert.mmbt.transformer.embeddings.word_embeddings.weight
    from mmf_amd.modules.encoders import FinetuneFasterRcnnFpnFc7, MultiModalEncoderBase
    from mmf_amd.modules.hf_layers import BertModelJit
    mc = mmbt_model_config(cfg, modal_encoder=dict(type="finetune_faster_rcnn_fpn_fc7", params=dict(
        in_dim=cfg["modal_hidden_size"], out_dim=cfg["modal_hidden_size"], weights_file="nope_w.pkl", bias_file="nope_b.pkl")))
    mc["text_encoder"]["params"]["num_segments"] = 3
    with pytest.warns(UserWarning, match="random initialisation"):
        model = build_model(mc)
    base = model.model.bert
    assert isinstance(base, MultiModalEncoderBase) and isinstance(base.mmbt.transformer, BertModelJit)
    assert isinstance(base.mmbt.modal_encoder.encoder, FinetuneFasterRcnnFpnFc7)
    assert base.mmbt.transformer.embeddings.token_type_embeddings.weight.shape[0] == 3 and base.num_max_segment == 3
    assert "model.bert.mmbt.modal_encoder.encoder.lc.weight" in dict(model.named_parameters())


def test_forward_without_native_gpu_fails_loudly():
    z, case, cfg, sd, sample = load_mmbt_case()
    model = build_mmbt(cfg, sd, O.SHARED, device="cpu")
    from mmf_amd.common.sample import SampleList
    with pytest.raises(Exception):
        model(SampleList(dict(sample)))


def test_freeze_flags():
    z, case, cfg, sd, sample = load_mmbt_case()
    model = build_model(mmbt_model_config(cfg, freeze_text=True))
    assert not any(p.requires_grad for p in model.model.bert.mmbt.transformer.parameters())
    assert model.model.bert.mmbt.modal_encoder.proj_embeddings.weight.requires_grad
    assert all(p.requires_grad for p in model.model.classifier.parameters())


def test_decoder_mode_hands_the_attention_a_causal_per_query_mask():
    """`is_decoder: true` (mmbt.py:244-272): the dry run shows every layer's attention launched with a materialised per-(query, key) mask, and the mask
    itself is the padding mask times key <= query."""
    from tests import native_stub
    from mmf_amd.common.sample import SampleList
    z, case, cfg, sd, sample = load_mmbt_case("mmbt_decoder64")
    model = build_mmbt(cfg, sd, O.SHARED, device="cpu")
    model.eval()
    seen = {}
    enc = model.model.bert.mmbt.transformer.encoder
    hook = enc.register_forward_pre_hook(lambda m, a: seen.update(mask=a[1]))
    with native_stub.installed() as calls:
        model(SampleList(dict(sample)))
    hook.remove()
    attn = [c for c in calls if c[0] == "attention_fwd"]
    assert len(attn) == cfg["num_hidden_layers"] and all(c[-1] == "per-query mask" for c in attn)
    m = seen["mask"]
    B, S = m.shape[0], m.shape[-1]
    assert tuple(m.shape) == (B, 1, S, S)
    vis = m[:, 0] == 0
    L = S - sample["input_ids"].shape[1]
    am = torch.cat([torch.ones(B, L, dtype=torch.long), torch.cat([sample["input_mask"][:, 1:], torch.zeros(B, 1, dtype=torch.long)], 1)], 1)      # (the end token shifts the text left, mmbt.py:346-363)
    want = torch.tril(torch.ones(S, S, dtype=torch.bool))[None] & (am[:, None, :] != 0)
    assert torch.equal(vis, want)
