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


def test_encoder_takes_a_materialised_per_query_mask():
    """`BertEncoderJit.forward` with the [B, 1, S, S] additive mask the reference's MMT builds (mmf/models/m4c.py:424-440; any mask
    `attention_scores + attention_mask` broadcasts, mmf/modules/hf_layers.py:187-190): outputs and every gradient are bit-identical to
    the unmaterialised form of the same mask (key mask + causal tail), in train mode with dropout."""
    import torch
    from transformers import BertConfig
    from mmf_amd import functional as Fn
    from mmf_amd.modules.hf_layers import BertEncoderJit, init_bert_weights
    B, S, H, tail = 2, 72, 128, 9
    cfg = BertConfig(hidden_size=H, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2, hidden_dropout_prob=0.1,
                     attention_probs_dropout_prob=0.1)
    torch.manual_seed(5)
    enc = BertEncoderJit(cfg)
    enc.apply(init_bert_weights)
    enc = enc.cuda().train()
    x0 = torch.randn(B, S, H, device="cuda")
    key = torch.zeros(B, S, device="cuda"); key[1, S - tail - 6:S - tail] = -10000.0
    q = torch.arange(S, device="cuda")[:, None]; k = torch.arange(S, device="cuda")[None, :]
    c0 = S - tail
    full = key[:, None, :].expand(B, S, S).clone()
    full[:, (k >= c0) & ~((q >= c0) & (k <= q))] = -10000.0
    full[:, (k >= c0) & (q >= c0) & (k <= q)] = 0.0
    res = []
    for mask in (Fn.PrefixLMMask(key.view(B, 1, 1, S), tail), full.view(B, 1, S, S)):
        enc.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        torch.manual_seed(99)       # eager dropout keys come from the device generator's seed and offset: the same masks in both runs
        out = enc(x, mask)[0]
        out.float().square().sum().backward()
        res.append((out.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in enc.named_parameters()}))
    assert torch.isfinite(res[0][0].float()).all()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for n in res[0][2]:
        assert torch.equal(res[0][2][n], res[1][2][n]), n
