"""Cutting the text columns no sample of a batch uses (mmf_amd/common/prefetch.py::trim_text_padding) changes no result: shown here on the CPU oracle
(the restatement of mmf/models/visual_bert.py pinned against the real reference) — scores, loss and every parameter gradient of the padded batch equal
those of the trimmed batch.  The reference masks padded keys and never compacts (visual_bert.py:94-106); this is the identity that lets the MI355X path
skip ~ 90 % of a real VQA batch's text rows."""
import torch

from mmf_amd.common.prefetch import DevicePrefetcher, trim_text_padding, used_text_length
from mmf_amd.common.sample import SampleList
from oracle import visual_bert_oracle as O
from tests.golden_utils import load_case


def padded_sample(lengths=(5, 3, 7), total=24):
    z, case, cfg, sd, sample = load_case("small64")
    B = sample["input_ids"].shape[0]
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(1, cfg["vocab_size"], (B, total), generator=g)
    lens = torch.tensor(lengths)
    mask = (torch.arange(total)[None, :] < lens[:, None]).long()
    s = dict(sample)
    s["input_ids"], s["input_mask"], s["segment_ids"] = ids * mask, mask, torch.zeros_like(mask)
    return cfg, sd, s


def test_used_text_length_and_fields():
    cfg, sd, s = padded_sample((5, 3, 7), 24)
    assert used_text_length(s["input_mask"], 8) == 8
    assert used_text_length(s["input_mask"], 4) == 8
    assert used_text_length(s["input_mask"], 1) == 7
    assert used_text_length(s["input_mask"], 16) == 16
    assert used_text_length(torch.ones(2, 24, dtype=torch.long), 8) == 24
    assert used_text_length(torch.zeros(2, 24, dtype=torch.long), 8) == 8          # (an all-masked batch keeps one bucket)
    hole = s["input_mask"].clone(); hole[1, 19] = 1                                    # not a prefix mask: the last USED column decides
    assert used_text_length(hole, 8) == 24 and used_text_length(hole, 1) == 20
    sl = SampleList(dict(s, lm_label_ids=torch.full_like(s["input_mask"], -1)))
    t = trim_text_padding(sl, 8)
    assert t is not sl and sl["input_ids"].shape == (3, 24)                            # the source batch is untouched
    for f in ("input_ids", "input_mask", "segment_ids", "lm_label_ids"):
        assert t[f].shape == (3, 8) and t[f].is_contiguous() and torch.equal(t[f], sl[f][:, :8]), f
    assert t["image_feature_0"] is sl["image_feature_0"] and t["targets"] is sl["targets"]
    assert t["dataset_name"] == "vqa2" and t.get_batch_size() == 3
    assert torch.equal(t["image_info_0"]["max_features"], sl["image_info_0"]["max_features"])
    full = SampleList(dict(s, input_mask=torch.ones_like(s["input_mask"])))
    assert trim_text_padding(full, 8) is full                                          # nothing to cut: the same object


def test_oracle_scores_loss_and_gradients_do_not_change():
    cfg, sd, s = padded_sample((5, 3, 7), 24)
    t = trim_text_padding(SampleList(s), 8)
    res = []
    for batch in (s, dict(t)):
        sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        out = O.train_step_loss(sdr, cfg, batch, train=False)
        loss = list(out["losses"].values())[0]
        loss.backward()
        res.append((out["scores"].detach(), loss.detach(), {k: v.grad for k, v in sdr.items()}))
    (s0, l0, g0), (s1, l1, g1) = res
    assert torch.allclose(s0, s1, rtol=1e-5, atol=1e-6)
    assert abs(float(l0) - float(l1)) <= 1e-6 * abs(float(l0))
    checked = 0
    for k in g0:
        if g0[k] is None:
            assert g1[k] is None, k
            continue
        if k.endswith("self.key.bias"):      # identically zero in exact arithmetic (softmax shift invariance): rounding noise on both sides
            continue
        e = float((g0[k] - g1[k]).double().norm() / (g0[k].double().norm() + 1e-30))
        assert e <= 1e-5, (k, e)             # fp32 summation order only (fewer zero terms in the softmax sums and the weight-gradient rows)
        checked += 1
    assert checked > 30
    # the position rows beyond the last used column receive exactly nothing in the untrimmed run either
    assert float(g0["bert.embeddings.position_embeddings.weight"][8:24].abs().max()) == 0.0


def test_prefetcher_trims_on_the_host():
    cfg, sd, s = padded_sample((5, 3, 7), 24)
    full = SampleList(dict(s, input_mask=torch.ones_like(s["input_mask"])))
    out = list(DevicePrefetcher([SampleList(s), full], device="cpu", trim_text_padding=8))
    assert out[0]["input_ids"].shape == (3, 8) and out[1]["input_ids"].shape == (3, 24)
    out = list(DevicePrefetcher([SampleList(s)], device="cpu"))
    assert out[0]["input_ids"].shape == (3, 24)


def _padded(sample, lengths, total, vocab):
    g = torch.Generator().manual_seed(11)
    B = sample["input_ids"].shape[0]
    lens = torch.tensor(lengths[:B])
    mask = (torch.arange(total)[None, :] < lens[:, None]).long()
    s = dict(sample)
    s["input_ids"] = torch.randint(1, vocab, (B, total), generator=g) * mask
    s["input_mask"], s["segment_ids"] = mask, torch.zeros_like(mask)
    return s


def _loss_and_grads(fn, sd, cfg, batch):
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = fn(sdr, cfg, batch, train=False)
    loss = sum(v.sum() for v in out["losses"].values())
    loss.backward()
    return out["scores"].detach(), float(loss), {k: v.grad for k, v in sdr.items()}


def _same(res0, res1):
    (s0, l0, g0), (s1, l1, g1) = res0, res1
    assert torch.allclose(s0, s1, rtol=1e-5, atol=1e-6)
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    n = 0
    for k in g0:
        if g0[k] is None or float(g0[k].abs().max()) == 0.0:
            assert g1[k] is None or float(g1[k].abs().max()) == 0.0, k
            continue
        if k.endswith(("key.bias", "key1.bias", "key2.bias")):      # identically zero in exact arithmetic (softmax shift invariance)
            continue
        e = float((g0[k] - g1[k]).double().norm() / g0[k].double().norm())
        assert e <= 1e-5, (k, e)
        n += 1
    return n


def test_vilbert_oracle_does_not_change():
    """Two streams + co-attention (mmf/models/vilbert.py:1092-1198): the text mask reaches the text stream's self-attention and the image -> text
    co-attention as an additive -10000; the pooled text output is position 0."""
    from oracle import vilbert_oracle as OV
    from tests.golden_utils import load_vilbert_case
    z, case, cfg, sd, sample = load_vilbert_case("vilbert_small")
    s = _padded(sample, (5, 3, 7), 24, cfg["vocab_size"])
    t = trim_text_padding(SampleList(s), 8)
    assert t["input_ids"].shape == (3, 8)

    def fn(sdr, cfg_, batch, train=False):
        out = OV.vilbert_forward(sdr, cfg_, batch, train=train)
        return {"scores": out["scores"], "losses": {"l": O.logit_bce(out["scores"], batch["targets"])}}
    assert _same(_loss_and_grads(fn, sd, cfg, s), _loss_and_grads(fn, sd, cfg, dict(t))) > 60


def test_mmbt_oracle_does_not_change():
    """MMBT (mmf/models/mmbt.py:244-272): [start] features [end] text; the text sits at the END of the joint sequence, its position ids are arange(T)."""
    from oracle import mmbt_oracle as OM
    from tests.golden_utils import load_mmbt_case
    z, case, cfg, sd, sample = load_mmbt_case("mmbt_small64")
    s = _padded(sample, (5, 3, 7, 8), 24, cfg["vocab_size"])
    t = trim_text_padding(SampleList(s), 8)
    assert t["input_ids"].shape == (4, 8)

    def fn(sdr, cfg_, batch, train=False):
        out = OM.mmbt_forward(sdr, cfg_, batch, train=train)
        return {"scores": out["scores"], "losses": {"l": OM.cross_entropy(out["scores"], batch["targets"])}}
    assert _same(_loss_and_grads(fn, sd, cfg, s), _loss_and_grads(fn, sd, cfg, dict(t))) > 30
