"""The call forms of /root/reference/tests/models/test_vilbert.py:55-137 against `mmf_amd.models.vilbert` (dry run, tests/native_stub.py): the
reference's tests build the registered model from its config and call `model.model(...)` with KEYWORD arguments — a single sample, ten regions
whose attention mask is all zero, `masked_lm_labels` / `image_label` filled with -1 (the latter as a FLOAT tensor) — and read `scores` /
`masked_lm_loss` from the output.  (There the eager output is compared with the scripted one: the `-m gpu` tests do that on hardware.)"""
import torch

from tests import golden_utils as G, model_utils as MU, native_stub

BERT_VOCAB_SIZE = 30255


def _inputs(cfg, num_bbox_per_image=10, T=32):
    g = torch.Generator().manual_seed(0)
    return dict(
        input_ids=torch.randint(low=0, high=min(BERT_VOCAB_SIZE, cfg["vocab_size"]), size=(1, T), generator=g).long(),
        attention_mask=torch.ones((1, T)).long(),
        token_type_ids=torch.zeros(1, T).long(),
        image_feature=torch.rand((1, num_bbox_per_image, cfg["v_feature_size"]), generator=g).float(),
        image_attention_mask=torch.zeros((1, num_bbox_per_image)).long(),
        image_location=torch.rand((1, num_bbox_per_image, 5), generator=g).float(),
    )


def test_finetune_model_keyword_call():
    z, case, cfg, sd, sample = G.load_vilbert_case()
    model = MU.build_vilbert(cfg, sd, device="cpu")
    model.eval()
    kw = _inputs(cfg, T=32)           # (the fixture's position table holds 40 rows; the reference's test uses 128 of BERT-base's 512)
    with native_stub.installed(), torch.no_grad():
        out = model.model(input_ids=kw["input_ids"], image_feature=kw["image_feature"], image_location=kw["image_location"],
                          token_type_ids=kw["token_type_ids"], attention_mask=kw["attention_mask"], image_attention_mask=kw["image_attention_mask"])
    assert out["scores"].shape == (1, cfg["num_labels"])


def test_pretrained_model_keyword_call_with_nothing_to_predict():
    z, case, cfg, sd, sample = G.load_vilbert_pretraining_case(0)
    model = MU.build_vilbert_pretraining(cfg, sd, device="cpu")
    model.eval()
    T = 32
    kw = _inputs(cfg, T=T)
    masked_lm_labels = torch.zeros((1, T), dtype=torch.long).fill_(-1)
    image_target = torch.zeros(1, 10, cfg["v_target_size"])
    image_label = torch.ones(1, 10).fill_(-1)                      # float, as the reference's test builds it
    with native_stub.installed(), torch.no_grad():
        out = model.model(input_ids=kw["input_ids"], image_feature=kw["image_feature"], image_location=kw["image_location"],
                          token_type_ids=kw["token_type_ids"], attention_mask=kw["attention_mask"], image_attention_mask=kw["image_attention_mask"],
                          masked_lm_labels=masked_lm_labels, image_label=image_label, image_target=image_target)
    assert "masked_lm_loss" in out and "masked_img_loss" in out
    assert out["masked_lm_loss"].numel() == 1 and out["masked_img_loss"].numel() == 1
