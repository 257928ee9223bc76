"""VisualBERT's masked-LM pretraining head on the GPU (mmf/models/visual_bert.py:160-281): the vocabulary-sized cross-entropy
kernels against PyTorch, the HIP model against the fixture produced by the real reference (logits, loss, every gradient incl. the
tied word-embedding table, which collects the gather AND the decoder gradient), the reference's own NaN test
(tests/models/test_visual_bert.py:71-98), and the fp32-accurate path at 1e-3."""
import math

import numpy as np
import pytest
import torch

import mmf_amd
from mmf_amd import _native as nat
from mmf_amd import functional as Fn
from mmf_amd.common.sample import SampleList
from tests import golden_utils as G
from tests.model_utils import build_visual_bert_pretraining, sample_to

pytestmark = pytest.mark.gpu
TOL = 5e-2


def rel_err(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.mark.parametrize("R,C", [(7, 211), (300, 30522), (64, 1000)])
def test_vocab_cross_entropy_kernels_match_torch(R, C):
    g = torch.Generator().manual_seed(R + C)
    logits = torch.randn(R, C, generator=g) * 3.0
    labels = torch.randint(0, C, (R,), generator=g)
    labels[torch.rand(R, generator=g) < 0.6] = -1
    labels[0] = C - 1
    x = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(x, labels, ignore_index=-1)
    ref.backward()
    ld = logits.cuda(); lab = labels.cuda()
    lse = torch.empty(R, device="cuda"); rowloss = torch.empty(R, device="cuda"); loss = torch.empty(1, device="cuda"); count = torch.empty(1, device="cuda")
    nat.vocab_cross_entropy_fwd(ld, lab, lse, rowloss, loss, count, R, C, -1)
    assert int(count.item()) == int((labels != -1).sum())
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    keep = labels != -1
    torch.testing.assert_close(lse.cpu()[keep], torch.logsumexp(logits[keep].double(), 1).float(), rtol=1e-5, atol=1e-5)
    ldd = (C + 7) // 8 * 8
    d = torch.full((R, ldd), 9.0, dtype=torch.bfloat16, device="cuda")
    gl = torch.tensor([2.0], device="cuda")
    nat.vocab_cross_entropy_bwd(ld, lab, lse, count, gl, d, ldd, R, C, -1)
    got = d.float().cpu()
    assert bool((got[:, C:] == 0).all()) and bool((got[~keep] == 0).all())
    want = 2.0 * x.grad
    assert rel_err(got[:, :C], want) <= 6e-3                                   # bf16 rounding of the stored gradient
    assert not nat.take_index_error()
    # every label ignored: NaN like torch; a label outside the vocabulary raises the index-error flag
    nat.vocab_cross_entropy_fwd(ld, torch.full((R,), -1, dtype=torch.int64, device="cuda"), lse, rowloss, loss, count, R, C, -1)
    assert torch.isnan(loss).item() and count.item() == 0
    bad = lab.clone(); bad[0] = C + 5
    nat.vocab_cross_entropy_fwd(ld, bad, lse, rowloss, loss, count, R, C, -1)
    assert nat.take_index_error()


def test_golden_pretraining_forward_loss_and_gradients():
    z, case, cfg, sd, sample = G.load_pretraining_case()
    model = build_visual_bert_pretraining(cfg, sd, output_hidden_states=True)
    model.eval()
    out = model(SampleList(sample_to(sample, "cuda")))
    assert out["logits"].shape == z["logits"].shape and out["logits"].dtype == torch.float32
    np.testing.assert_allclose(out["logits"].detach().cpu().numpy(), z["logits"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(out["sequence_output"].detach().float().cpu().numpy(), z["sequence_output"], rtol=TOL, atol=TOL)
    (key, loss), = out["losses"].items()
    assert key == str(z["loss_key"]) == "coco/train/masked_lm_loss" and "masked_lm_loss" not in out and out["loss"] is loss
    assert abs(loss.item() - float(z["loss"])) <= TOL * abs(float(z["loss"]))
    loss.backward()
    params = dict(model.named_parameters())
    bad = {}
    for gname, norm in zip(z["grad_names"], z["grad_norms"]):
        gname = str(gname)
        p = params[gname]
        if norm == 0.0:                    # pooler, next-sentence head: outside the loss in the reference too
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, gname
            continue
        assert p.grad is not None, gname
        if gname.endswith("self.key.bias"):
            continue
        e = abs(float(p.grad.double().norm()) - norm) / norm
        if e > TOL:
            bad[gname] = e
        full = "grad::" + gname
        if full in z.files:
            assert rel_err(p.grad, torch.from_numpy(z[full])) <= TOL, gname
    assert not bad, bad
    # the tied table: one Parameter, one gradient = embedding-gather gradient + decoder weight gradient
    assert model.model.cls.predictions.decoder.weight is model.model.bert.embeddings.word_embeddings.weight


def test_all_labels_ignored_gives_nan_like_the_reference_test():
    """tests/models/test_visual_bert.py:71-98 of the reference: lm_label_ids all -1 -> "random/test/masked_lm_loss" is NaN."""
    z, case, cfg, sd, sample = G.load_pretraining_case()
    model = build_visual_bert_pretraining(cfg, sd)
    model.eval()
    s = dict(sample, lm_label_ids=torch.full_like(sample["lm_label_ids"], -1), dataset_name="random", dataset_type="test")
    with torch.no_grad():
        out = model(SampleList(sample_to(s, "cuda")))
    assert "losses" in out and "random/test/masked_lm_loss" in out["losses"]
    assert torch.isnan(out["losses"]["random/test/masked_lm_loss"])


def test_pretraining_fp32_path_within_1e3_of_the_reference():
    z, case, cfg, sd, sample = G.load_pretraining_case()
    model = build_visual_bert_pretraining(cfg, sd)
    model.eval()
    with mmf_amd.fp32_inference():
        out = model(SampleList(sample_to(sample, "cuda")))
    e = np.abs(out["logits"].cpu().numpy() - z["logits"]).max()
    (key, loss), = out["losses"].items()
    assert e <= 1e-3 and abs(loss.item() - float(z["loss"])) <= 1e-3 * abs(float(z["loss"])), (e, loss.item())


def test_pretraining_train_step_at_bert_vocabulary_size():
    """One training step (dropout on, fused AdamW) with the real vocabulary (30522: ragged N for the decoder GEMM, K = 30522 for
    its input gradient, a [30522, H] weight gradient landing on the tied table): finite, and the loss goes down over a few steps."""
    from mmf_amd.common.registry import registry
    from mmf_amd.utils.configuration import Config
    z, case, cfg, sd, sample = G.load_pretraining_case()
    cfg = dict(cfg, vocab_size=30522)
    model = build_visual_bert_pretraining(cfg, None)
    model.train()
    g = torch.Generator().manual_seed(3)
    B, T = sample["input_ids"].shape
    s = dict(sample, input_ids=torch.randint(1, 30522, (B, T), generator=g))
    lm = torch.where(torch.rand(B, T, generator=g) < 0.3, s["input_ids"], torch.full((B, T), -1))
    lm[:, 1] = s["input_ids"][:, 1]
    s["lm_label_ids"] = lm
    full = Config(model="visual_bert", optimizer=dict(params=dict(lr=1e-3)), model_config={"visual_bert": model.config})
    opt = registry.get_optimizer_class("adam_w")(model.get_optimizer_parameters(full), lr=1e-3, eps=1e-8)
    batch = SampleList(sample_to(s, "cuda"))
    losses = []
    for _ in range(6):
        out = model(batch)
        (key, loss), = out["losses"].items()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0] - 0.5, losses
    w = model.model.bert.embeddings.word_embeddings.weight
    assert w.grad is not None and w.grad.shape == (30522, cfg["hidden_size"]) and bool(torch.isfinite(w.grad).all())


def test_mmbt_pretraining_golden_forward_loss_gradients_and_state_dict():
    """MMBTForPreTraining (mmf/models/mmbt.py:447-523) against the fixture recorded from the reference's own forward: logits over
    all positions, masked-LM loss over the text positions, every gradient; state-dict keys equal the reference module tree's."""
    from oracle.mmbt_oracle import SHARED
    from tests.model_utils import build_mmbt_pretraining
    z, case, cfg, sd, sample = G.load_mmbt_pretraining_case()
    model = build_mmbt_pretraining(cfg, sd, SHARED)
    assert sorted(model.state_dict().keys()) == sorted(str(k) for k in z["state_dict_keys"] if not str(k).endswith("position_ids"))
    model.eval()
    out = model(SampleList(sample_to(sample, "cuda")))
    np.testing.assert_allclose(out["logits"].detach().cpu().numpy(), z["logits"], rtol=TOL, atol=TOL)
    (key, loss), = out["losses"].items()
    assert key == str(z["loss_key"])
    assert abs(loss.item() - float(z["loss"])) <= TOL * abs(float(z["loss"]))
    loss.backward()
    params = dict(model.named_parameters())
    for gname, norm in zip(z["grad_names"], z["grad_norms"]):
        gname = str(gname)
        name = "model." + SHARED.get(gname[len("model."):], gname[len("model."):])
        p = params[name]
        if norm == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, gname
            continue
        if gname.endswith("self.key.bias"):
            continue
        assert p.grad is not None and abs(float(p.grad.double().norm()) - norm) <= TOL * norm, gname
        full = "grad::" + gname
        if full in z.files:
            assert rel_err(p.grad, torch.from_numpy(z[full])) <= TOL, gname
    with mmf_amd.fp32_inference():
        out32 = model(SampleList(sample_to(sample, "cuda")))
    assert np.abs(out32["logits"].cpu().numpy() - z["logits"]).max() <= 1e-3


@pytest.mark.parametrize("R,C", [(21, 53), (100, 1601)])
def test_soft_target_kl_kernels_match_torch(R, C):
    g = torch.Generator().manual_seed(R * C)
    logits = torch.randn(R, C, generator=g) * 2.0
    raw = torch.rand(R, C, generator=g)
    raw = torch.where(raw < 0.5, torch.zeros_like(raw), raw)
    raw[:, 0] += 1e-3
    target = raw / raw.sum(1, keepdim=True)
    label = torch.randint(-1, 2, (R,), generator=g)
    label[0] = 1
    x = logits.clone().requires_grad_(True)
    kl = torch.nn.functional.kl_div(torch.log_softmax(x, 1), target, reduction="none")
    ref = (kl * (label == 1).unsqueeze(1).float()).sum() / (label == 1).sum()
    ref.backward()
    xd, td, ld_ = logits.cuda(), target.cuda(), label.cuda()
    lse = torch.empty(R, device="cuda"); tsum = torch.empty(R, device="cuda"); rowloss = torch.empty(R, device="cuda")
    loss = torch.empty(1, device="cuda"); count = torch.empty(1, device="cuda")
    nat.soft_target_kl_fwd(xd, td, ld_, lse, tsum, rowloss, loss, count, R, C)
    assert int(count.item()) == int((label == 1).sum()) and abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    ldd = (C + 7) // 8 * 8
    d = torch.full((R, ldd), 9.0, dtype=torch.bfloat16, device="cuda")
    nat.soft_target_kl_bwd(xd, td, ld_, lse, tsum, count, torch.tensor([1.5], device="cuda"), d, ldd, R, C)
    got = d.float().cpu()
    assert bool((got[:, C:] == 0).all()) and bool((got[label != 1] == 0).all())
    assert rel_err(got[:, :C], 1.5 * x.grad) <= 6e-3


@pytest.mark.parametrize("visual_target", [0, 1, 2])        # 2: NCE, the sampled negatives replayed from the fixture
def test_vilbert_pretraining_golden_losses_gradients_and_state_dict(visual_target):
    """ViLBERTForPretraining (mmf/models/vilbert.py:1054-1240; visual_target 0: KL against the detector's class distribution, 1: masked-region
    regression with nn.MSELoss, :1139-1148 — round 3) against the reference's own run: masked-LM and masked-region losses (both shaped [1],
    keyed like the reference), every gradient, state-dict keys."""
    from tests.model_utils import build_vilbert_pretraining
    z, case, cfg, sd, sample = G.load_vilbert_pretraining_case(visual_target)
    over = dict(num_negative=cfg["num_negative"]) if visual_target == 2 else {}
    model = build_vilbert_pretraining(cfg, sd, visual_target=visual_target, **over)
    assert sorted(model.state_dict().keys()) == sorted(str(k) for k in z["state_dict_keys"])
    model.eval()
    sample = {k: v for k, v in sample.items() if not k.startswith("_")}
    if visual_target == 2:
        with G.recorded_random(z):        # Tensor.random_ replays the reference run's draws: the same negatives
            out = model(SampleList(sample_to(sample, "cuda")))
    else:
        out = model(SampleList(sample_to(sample, "cuda")))
    ref = dict(zip((str(k) for k in z["loss_keys"]), z["loss_values"]))
    assert set(out["losses"]) == set(ref)
    for k, v in out["losses"].items():
        assert tuple(v.shape) == (1,) and abs(v.item() - ref[k]) <= TOL * abs(ref[k]), (k, v.item(), ref[k])
    sum(v.sum() for v in out["losses"].values()).backward()
    params = dict(model.named_parameters())
    bad = {}
    for gname, norm in zip(z["grad_names"], z["grad_norms"]):
        gname = str(gname)
        p = params[gname]
        if norm == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, gname
            continue
        if gname.endswith(".key.bias") or gname.endswith("key1.bias") or gname.endswith("key2.bias"):
            continue
        assert p.grad is not None, gname
        e = abs(float(p.grad.double().norm()) - norm) / norm
        if e > TOL:
            bad[gname] = e
        full = "grad::" + gname
        if full in z.files and norm > 1e-6:
            e2 = rel_err(p.grad, torch.from_numpy(z[full]))
            # query-bias gradients of the near-uniform attention of this toy fixture are differences of nearly equal terms (DESIGN §2,
            # conditioning fact (ii)): element-wise they sit at the bf16 bound itself (observed 5.2e-2 on the NCE fixture), their norms within it
            if e2 > (1.6 * TOL if gname.endswith("attention.self.query.bias") else TOL):
                bad[gname + " (full)"] = e2
    assert not bad, bad


def test_nce_kernels_match_torch():
    """mmf_nce_fwd / mmf_nce_bwd (ViLBERT `visual_target: 2`, vilbert.py:1205-1227) against torch: bmm scores against the own target and the
    gathered negatives, CrossEntropyLoss on class 0 over the labelled regions; gradient with respect to the prediction as the zero-padded bf16
    operand; no labelled region gives NaN."""
    from mmf_amd import _native as nat
    M, N, K = 37, 52, 10
    g = torch.Generator().manual_seed(3)
    pred = torch.randn(M, N, generator=g) * 0.5
    target = torch.randn(M, N, generator=g)
    neg = torch.randint(0, M, (M, K), generator=g)
    label = (torch.rand(M, generator=g) < 0.4).long()
    label[3] = 1; label[5] = -1
    pd = pred.double().requires_grad_(True)
    pick = label == 1
    sample = torch.cat((target.double()[pick].unsqueeze(1), target.double()[neg[pick]]), dim=1)
    score = torch.bmm(sample, pd[pick].unsqueeze(2)).squeeze(2)
    ref = torch.nn.functional.cross_entropy(score, torch.zeros(score.size(0), dtype=torch.long))
    (1.5 * ref).backward()
    dev = "cuda"
    scores = torch.empty(M, K + 1, device=dev); lse = torch.empty(M, device=dev); rowloss = torch.empty(M, device=dev)
    loss = torch.empty(1, device=dev); count = torch.empty(1, device=dev)
    nat.nce_fwd(pred.to(dev), target.to(dev), neg.to(dev), label.to(dev), scores, lse, rowloss, loss, count, M, N, K)
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item()) and count.item() == float(pick.sum())
    ldd = (N + 7) // 8 * 8
    d = torch.full((M, ldd), float("nan"), dtype=torch.bfloat16, device=dev)
    nat.nce_bwd(target.to(dev), neg.to(dev), label.to(dev), scores, lse, count, torch.full((1,), 1.5, device=dev), d, ldd, M, N, K)
    got = d.float().cpu()
    assert bool((got[:, N:] == 0).all()) and bool((got[~pick] == 0).all())
    assert rel_err(got[:, :N], pd.grad) <= 6e-3
    none = torch.zeros(M, dtype=torch.long, device=dev)
    nat.nce_fwd(pred.to(dev), target.to(dev), neg.to(dev), none, scores, lse, rowloss, loss, count, M, N, K)
    assert math.isnan(loss.item())
