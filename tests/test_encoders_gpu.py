"""tests/modules/test_encoders.py:56-71 of the reference (`test_transformer_encoder_forward`), on the HIP-backed encoder: the registered
`"transformer"` encoder returns the pooled output by default and the sequence with `return_sequence=True`; and ViLBERT's pretraining
model with every label ignored gives a NaN masked-LM loss (tests/models/test_vilbert.py:55-101)."""
import pytest
import torch

from mmf_amd.modules import encoders

pytestmark = pytest.mark.gpu


def test_transformer_encoder_forward():
    encoder = encoders.TransformerEncoder.from_params().cuda()
    encoder.eval()
    assert encoder.embeddings.word_embeddings.weight.size(1) == 768
    assert encoder.embeddings.word_embeddings.weight.size(0) == 30522
    text_ids = torch.randint(encoder.embeddings.word_embeddings.weight.size(0), (2, 16)).cuda()
    with torch.no_grad():
        text_embeddings_cls = encoder(text_ids)
        assert text_embeddings_cls.dim() == 2 and list(text_embeddings_cls.size()) == [2, 768]
        text_embeddings = encoder(text_ids, return_sequence=True)
        assert text_embeddings.dim() == 3 and list(text_embeddings.size()) == [2, 16, 768]
    assert bool(torch.isfinite(text_embeddings.float()).all()) and float(text_embeddings_cls.float().abs().max()) <= 1.0     # tanh pooler


def test_vilbert_pretrained_model_all_labels_ignored():
    from tests import golden_utils as G
    from tests.model_utils import build_vilbert_pretraining
    z, case, cfg, sd, sample = G.load_vilbert_pretraining_case()
    model = build_vilbert_pretraining(cfg, sd)
    model.eval()
    B, R, T = sample["image_feature_0"].shape[0], sample["image_feature_0"].shape[1], sample["input_ids"].shape[1]
    with torch.no_grad():
        out = model.model(
            input_ids=sample["input_ids"].cuda(), image_feature=sample["image_feature_0"].cuda(),
            image_location=sample["image_info_0"]["bbox"].cuda(), token_type_ids=sample["segment_ids"].cuda(),
            attention_mask=sample["input_mask"].cuda(), image_attention_mask=torch.ones(B, R, dtype=torch.long).cuda(),
            masked_lm_labels=torch.full((B, T), -1, dtype=torch.long).cuda(),
            image_label=torch.full((B, R), -1, dtype=torch.long).cuda(),
            image_target=torch.zeros(B, R, cfg["v_target_size"]).cuda())
    assert tuple(out["masked_lm_loss"].shape) == (1,) and bool(torch.isnan(out["masked_lm_loss"]))
    assert tuple(out["masked_img_loss"].shape) == (1,)
