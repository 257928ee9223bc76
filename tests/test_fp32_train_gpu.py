"""fp32 training kernels (mmf_amd/csrc/fp32_train.hip, the attention backward in fp32_path.hip) against float64 PyTorch autograd of the
same operations, and the fp32 VisualBERT training step (`mmf_amd.fp32_training()`) against the real reference's fixture and the CPU
oracle: every parameter gradient within north_star's fp32 bound (1e-3)."""
import math

import numpy as np
import pytest
import torch

import mmf_amd
from mmf_amd import _native as nat
from mmf_amd.common.sample import SampleList
from oracle import visual_bert_oracle as O
from tests import golden_utils as G
from tests.model_utils import build_visual_bert, sample_to

pytestmark = pytest.mark.gpu
TOL_FP32 = 1e-3
KERNEL_TOL = 5e-5


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def _ref_attention(q, k, v, ext, scale, B, heads, Sq, Sk, hd, keep=None):
    qd = q.view(B, Sq, heads, hd).transpose(1, 2)
    kd = k.view(B, Sk, heads, hd).transpose(1, 2)
    vd = v.view(B, Sk, heads, hd).transpose(1, 2)
    p = torch.softmax(qd @ kd.transpose(-1, -2) * scale + ext, -1)
    if keep is not None:
        p = p * keep
    return (p @ vd).transpose(1, 2).reshape(B * Sq, heads * hd)


@pytest.mark.parametrize("B,heads,Sq,Sk,hd,tail", [(2, 2, 24, 24, 64, 0), (2, 12, 228, 228, 64, 0), (3, 2, 40, 150, 64, 0), (2, 2, 70, 70, 64, 12),
                                                    (2, 8, 101, 128, 128, 0), (2, 4, 128, 37, 128, 0), (1, 2, 256, 256, 64, 0),
                                                    (2, 3, 356, 356, 64, 0), (1, 2, 512, 512, 64, 0), (2, 2, 40, 300, 64, 0), (2, 2, 300, 130, 64, 0),
                                                    (2, 2, 300, 300, 64, 12), (2, 2, 200, 200, 128, 0), (1, 2, 256, 129, 128, 0)])
def test_attention_f32_backward_matches_float64_autograd(B, heads, Sq, Sk, hd, tail):
    H = heads * hd
    scale = 1.0 / math.sqrt(hd)
    q, k, v, do = _rand(B * Sq, H, seed=1), _rand(B * Sk, H, seed=2), _rand(B * Sk, H, seed=3), _rand(B * Sq, H, seed=4)
    key_mask = torch.ones(B, Sk)
    key_mask[0, Sk // 2: Sk // 2 + 5] = 0
    key_mask[-1, ::3] = 0
    key_mask[-1, 0] = 1
    if tail:
        key_mask[:, Sk - tail:] = 0
    ext = key_mask[:, None, None, :].repeat(1, 1, Sq, 1)
    if tail:
        ext[:, :, Sq - tail:, Sk - tail:] = torch.tril(torch.ones(tail, tail))
    ext = ((1.0 - ext) * -10000.0).double()
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = _ref_attention(qd, kd, vd, ext, scale, B, heads, Sq, Sk, hd)
    ref.backward(do.double())
    qc, kc, vc, doc = q.cuda(), k.cuda(), v.cuda(), do.cuda()
    mask = ((1.0 - key_mask) * -10000.0).cuda()
    out = torch.full((B * Sq, H), float("nan"), device="cuda")
    lse = torch.empty(B, heads, Sq, device="cuda")
    nat.attention_f32_fwd(qc, kc, vc, H, H, H, mask, out, H, B, heads, Sq, Sk, scale, head_dim=hd, causal_tail=tail, lse=lse)
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=KERNEL_TOL, atol=KERNEL_TOL)
    dq, dk, dv = (torch.full_like(t, float("nan")) for t in (qc, kc, vc))
    delta = torch.empty(B, heads, Sq, device="cuda")
    nat.attention_f32_bwd(qc, kc, vc, H, H, H, mask, out, H, lse, B, heads, Sq, Sk, scale, doc, dq, dk, dv, delta, head_dim=hd, causal_tail=tail)
    for name, got, want in (("dq", dq, qd.grad), ("dk", dk, kd.grad), ("dv", dv, vd.grad)):
        torch.testing.assert_close(got.cpu().double(), want, rtol=2 * KERNEL_TOL, atol=2 * KERNEL_TOL, msg=lambda m, n=name: n + ": " + m)


def test_attention_f32_dropout_forward_and_backward_use_the_same_mask():
    """Probability dropout: the mask is recovered from the forward itself (V = unit rows, 64 keys at a time), then forward and backward are
    checked against float64 autograd with THAT mask; keep rate ~ 1 - p; the same key reproduces the same mask."""
    B, heads, S, hd, p = 2, 2, 100, 64, 0.25
    H = heads * hd
    scale = 1.0 / math.sqrt(hd)
    q, k, v, do = _rand(B * S, H, seed=1), _rand(B * S, H, seed=2), _rand(B * S, H, seed=3), _rand(B * S, H, seed=4)
    drop = nat.drop_cfg(p, 4242)
    qc, kc, vc, doc = q.cuda(), k.cuda(), v.cuda(), do.cuda()

    def fwd(vv, dr):
        out = torch.empty(B * S, H, device="cuda"); lse = torch.empty(B, heads, S, device="cuda")
        nat.attention_f32_fwd(qc, kc, vv, H, H, H, None, out, H, B, heads, S, S, scale, lse=lse, drop=dr)
        return out, lse
    pd = torch.zeros(B, heads, S, S, dtype=torch.float64)
    for c0 in range(0, S, hd):
        ve = torch.zeros(B, S, heads, hd)
        for kk in range(c0, min(S, c0 + hd)):
            ve[:, kk, :, kk - c0] = 1.0
        o, _ = fwd(ve.reshape(B * S, H).cuda(), drop)
        pd[..., c0:c0 + hd] = o.cpu().double().view(B, S, heads, hd).transpose(1, 2)[..., : min(S, c0 + hd) - c0]
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    pfull = torch.softmax(qd.view(B, S, heads, hd).transpose(1, 2) @ kd.view(B, S, heads, hd).transpose(1, 2).transpose(-1, -2) * scale, -1)
    keep = torch.where(pfull.detach() > 1e-12, pd / pfull.detach(), torch.ones_like(pd))
    kept = keep > 0.5
    assert abs(float(kept.double().mean()) - (1 - p)) < 0.01
    torch.testing.assert_close(keep[kept], torch.full_like(keep[kept], 1.0 / (1 - p)), rtol=1e-4, atol=1e-4)
    keep = kept.double() / (1 - p)
    ref = _ref_attention(qd, kd, vd, 0.0, scale, B, heads, S, S, hd, keep=keep)
    ref.backward(do.double())
    out, lse = fwd(vc, drop)
    out2, _ = fwd(vc, drop)
    assert torch.equal(out, out2)
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=KERNEL_TOL, atol=KERNEL_TOL)
    dq, dk, dv = (torch.empty_like(t) for t in (qc, kc, vc))
    delta = torch.empty(B, heads, S, device="cuda")
    nat.attention_f32_bwd(qc, kc, vc, H, H, H, None, out, H, lse, B, heads, S, S, scale, doc, dq, dk, dv, delta, drop=drop)
    for name, got, want in (("dq", dq, qd.grad), ("dk", dk, kd.grad), ("dv", dv, vd.grad)):
        torch.testing.assert_close(got.cpu().double(), want, rtol=2 * KERNEL_TOL, atol=2 * KERNEL_TOL, msg=lambda m, n=name: n + ": " + m)


@pytest.mark.parametrize("rows,H", [(5, 128), (7296, 768), (300, 1024), (33, 2048)])
def test_layernorm_f32_backward(rows, H):
    x, dy = _rand(rows, H, seed=1, scale=2.0) + 0.3, _rand(rows, H, seed=2)
    gamma, beta = _rand(H, seed=3) * 0.1 + 1.0, _rand(H, seed=4) * 0.1
    xd, gd, bd = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (H,), gd, bd, 1e-12)
    ref.backward(dy.double())
    xc, gc = x.cuda(), gamma.cuda()
    y = torch.empty(rows, H, device="cuda"); mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    nat.layernorm_f32_fwd_stats(xc, gc, beta.cuda(), y, mean, rstd, rows, H, 1e-12)
    torch.testing.assert_close(y.cpu().double(), ref.detach(), rtol=KERNEL_TOL, atol=KERNEL_TOL)
    dx = torch.full((rows, H), float("nan"), device="cuda"); dg = torch.empty(H, device="cuda"); db = torch.empty(H, device="cuda")
    nat.layernorm_f32_bwd(dy.cuda(), xc, mean, rstd, gc, dx, dg, db, rows, H)
    torch.testing.assert_close(dx.cpu().double(), xd.grad, rtol=KERNEL_TOL, atol=KERNEL_TOL)
    tol = KERNEL_TOL * max(1.0, math.sqrt(rows / 100.0))
    torch.testing.assert_close(dg.cpu().double(), gd.grad, rtol=tol, atol=tol)
    torch.testing.assert_close(db.cpu().double(), bd.grad, rtol=tol, atol=tol)


def test_colsum_dropout_and_row_scatter_f32():
    rows, N = 7296, 3072
    x = _rand(rows, N, seed=1)
    out = torch.empty(N, device="cuda")
    nat.colsum_f32(x.cuda(), N, rows, N, out)
    torch.testing.assert_close(out.cpu().double(), x.double().sum(0), rtol=1e-4, atol=2e-4)
    nat.colsum_f32(x.cuda(), N, rows, N, out, accumulate=True)
    torch.testing.assert_close(out.cpu().double(), 2 * x.double().sum(0), rtol=1e-4, atol=4e-4)
    narrow = torch.empty(37, device="cuda")
    nat.colsum_f32(x.cuda()[:, 5:], N, 100, 37, narrow)                                          # a column slice of a wider buffer
    torch.testing.assert_close(narrow.cpu().double(), x.double()[:100, 5:42].sum(0), rtol=1e-4, atol=1e-4)
    drop = nat.drop_cfg(0.1, 77)
    xs = x.cuda()[:1000].contiguous()
    y1, y2 = torch.empty_like(xs), torch.empty_like(xs)
    nat.dropout_f32(xs, y1, drop); nat.dropout_f32(xs, y2, drop)
    assert torch.equal(y1, y2)
    kept = y1 != 0
    assert abs(float(kept.float().mean()) - 0.9) < 0.005
    torch.testing.assert_close(y1[kept], (xs / (1 - round(0.1 * 65536) / 65536))[kept], rtol=1e-5, atol=1e-6)
    # embedding backward: rows b * S + t (t < T) of a [B, S, H] gradient scattered by token id, the padding id dropped
    B, T, S, H, V = 4, 6, 10, 64, 9
    g = _rand(B * S, H, seed=5)
    ids = torch.randint(0, V, (B * T,), generator=torch.Generator().manual_seed(6))
    table = torch.zeros(V, H, device="cuda")
    nat.scatter_add_rows_f32(g.cuda(), H, B * T, H, ids.cuda(), table, H, grp=(T, S, 0), skip=0)
    ref = torch.zeros(V, H, dtype=torch.float64)
    src = g.double().view(B, S, H)[:, :T].reshape(B * T, H)
    ref.index_add_(0, ids, src)
    ref[0] = 0
    torch.testing.assert_close(table.cpu().double(), ref, rtol=1e-5, atol=1e-5)
    # pooling-gather backward: one row per sample at b * S + index[b]
    idx = torch.tensor([3, 0, 9, 5])
    gp = _rand(B, H, seed=7)
    full = torch.zeros(B * S, H, device="cuda")
    nat.scatter_add_rows_f32(gp.cuda(), H, B, H, idx.cuda(), full, H, dst_stride=S)
    want = torch.zeros(B, S, H); want[torch.arange(B), idx] = gp
    assert torch.equal(full.cpu(), want.view(B * S, H))


def _rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _check_all_gradients(model, sdr, tol):
    params = dict(model.named_parameters())
    errs = {}
    for k, v in sdr.items():
        p = params["model." + k]
        if v.grad is None or float(v.grad.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None and p.grad.dtype == torch.float32, k
        if k.endswith("self.key.bias"):
            # d/d(key bias) is identically zero in exact arithmetic (a per-query constant shift of the scores cancels in the softmax):
            # both sides hold rounding noise only
            qn = float(params["model." + k.replace("key.bias", "query.bias")].grad.double().norm())
            assert float(p.grad.double().norm()) <= tol * qn + 1e-6, k
            continue
        errs[k] = _rel(p.grad, v.grad)
    bad = {k: e for k, e in errs.items() if e > tol}
    assert not bad, bad
    return errs


def test_fp32_training_golden_small64_gradients_within_the_fp32_bound():
    """The reference's own fixture (tests/golden/make_golden.py ran /root/reference's VisualBERT + LogitBinaryCrossEntropy, forward and
    backward, in fp32): loss, scores, gradient norms and sums of every parameter, full gradients of the small ones."""
    z, case, cfg, sd, sample = G.load_case("small64")
    model = build_visual_bert(cfg, sd, output_hidden_states=True)
    model.eval()
    with mmf_amd.fp32_training():
        out = model(SampleList(sample_to(sample, "cuda")))
    assert out["scores"].dtype == torch.float32 and out["scores"].requires_grad
    np.testing.assert_allclose(out["scores"].detach().cpu().numpy(), z["scores"], rtol=TOL_FP32, atol=TOL_FP32)
    (key, loss), = out["losses"].items()
    assert key == "train/vqa2/logit_bce" and abs(loss.item() - float(z["loss"])) <= TOL_FP32 * abs(float(z["loss"]))
    loss.backward()
    params = dict(model.named_parameters())
    for gname, norm, gsum in zip(z["grad_names"], z["grad_norms"], z["grad_sums"]):
        gname = str(gname)
        p = params[gname]
        if norm == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, gname
            continue
        assert p.grad is not None, gname
        if gname.endswith("self.key.bias"):
            continue
        assert abs(float(p.grad.double().norm()) - norm) <= TOL_FP32 * norm, gname
        assert abs(float(p.grad.double().sum()) - gsum) <= TOL_FP32 * norm + 1e-7, gname
        full = "grad::" + gname
        if full in z.files:
            assert _rel(p.grad, torch.from_numpy(z[full])) <= TOL_FP32, gname


def test_fp32_training_full_config_every_gradient_matches_the_oracle():
    """VisualBERT-base VQA2 (12 layers, 128 tokens + 100 regions, ragged lengths), eval mode (dropout off), B = 4: every parameter
    gradient against the CPU oracle's autograd, relative L2 error <= 1e-3 (observed ~1e-6)."""
    cfg = dict(O.DEFAULT_CONFIG)
    cfg["num_hidden_layers"] = 12
    sd = O.init_state_dict(cfg, seed=7)
    g = torch.Generator().manual_seed(8)
    for k in sd:
        if k.endswith(".bias"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.02
        elif k.endswith("LayerNorm.weight"):
            sd[k] = 1.0 + torch.randn(sd[k].shape, generator=g) * 0.05
    sample = O.synthetic_batch(cfg, 4, seed=99)
    sample["input_mask"][1, 90:] = 0
    sample["image_info_0"]["max_features"][0] = 73
    model = build_visual_bert(cfg, sd)
    model.eval()
    with mmf_amd.fp32_training():
        out = model(SampleList(sample_to(sample, "cuda")))
    (key, loss), = out["losses"].items()
    loss.backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.visual_bert_forward(sdr, cfg, sample, train=False)
    ref_loss = O.logit_bce(ref["scores"], sample["targets"])
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= TOL_FP32 * abs(ref_loss.item())
    assert (out["scores"].detach().cpu() - ref["scores"].detach()).abs().max().item() <= TOL_FP32
    errs = _check_all_gradients(model, sdr, TOL_FP32)
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    json.dump({"parameters_compared": len(errs), "worst_rel_l2": worst, "loss": loss.item(), "oracle_loss": ref_loss.item()},
              open("gpurun_out/fp32_training_grad_errors.json", "w"), indent=1)


def test_fp32_training_step_with_dropout_and_fused_adamw():
    """Train mode (dropout 0.1 everywhere, the hash masks regenerated in the backward), a few steps of the fused AdamW on the fp32
    gradients: reproducible under torch.manual_seed (to the summation order of the embedding scatters), the loss on the fixed batch goes down, nothing is left in the bf16 routing afterwards."""
    from mmf_amd.modules.optimizers import AdamW
    z, case, cfg, sd, sample = G.load_case("small64")

    def run(steps):
        torch.manual_seed(1234)
        model = build_visual_bert(cfg, sd)
        model.train()
        opt = AdamW(model.parameters(), lr=2e-3, weight_decay=0.01)
        batch = SampleList(sample_to(sample, "cuda"))
        losses = []
        for _ in range(steps):
            opt.zero_grad()
            with mmf_amd.fp32_training():
                out = model(batch)
            (key, loss), = out["losses"].items()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        return losses, model
    a, model = run(6)
    b, _ = run(6)
    # same dropout masks, same arithmetic; only the embedding-table scatter adds (fp32 atomics) may reorder sums between runs
    assert a[0] == b[0] and all(abs(x - y) <= 1e-5 * abs(x) for x, y in zip(a, b)), (a, b)
    assert a[-1] < a[0], a
    model.eval()
    out = model(SampleList(sample_to(sample, "cuda")))           # the bf16 path still runs afterwards
    assert out["scores"].requires_grad


def test_fp32_training_nlvr2_head_golden():
    """`training_head_type: nlvr2` (two images per sample, BertPooler's tanh, the pairing of the two halves, cross_entropy) against the
    reference's fixture: loss, scores, every gradient norm / sum."""
    z, case, cfg, sd, sample = G.load_nlvr2_case()
    model = build_visual_bert(cfg, sd, training_head_type="nlvr2", pooler_strategy="default", losses=[dict(type="cross_entropy")])
    model.eval()
    with mmf_amd.fp32_training():
        out = model(SampleList(sample_to(sample, "cuda")))
    np.testing.assert_allclose(out["scores"].detach().cpu().numpy(), z["scores"], rtol=TOL_FP32, atol=TOL_FP32)
    (key, loss), = out["losses"].items()
    assert abs(loss.item() - float(z["loss"])) <= TOL_FP32 * abs(float(z["loss"]))
    loss.backward()
    params = dict(model.named_parameters())
    checked = 0
    for gname, norm, gsum in zip(z["grad_names"], z["grad_norms"], z["grad_sums"]):
        gname = str(gname)
        if norm == 0.0 or gname.endswith("self.key.bias"):
            continue
        p = params[gname]
        assert p.grad is not None, gname
        assert abs(float(p.grad.double().norm()) - norm) <= TOL_FP32 * norm, gname
        assert abs(float(p.grad.double().sum()) - gsum) <= TOL_FP32 * norm + 1e-7, gname
        checked += 1
    assert checked > 30


def test_fp32_training_refuses_what_it_does_not_build():
    """What mmf_amd.fp32_training() does not build raises instead of silently dropping to bf16: more positions than any attention kernel of the library
    takes (512 at head_dim 64, 256 at head_dim 128).  (ViLBERT's `in_batch_pairs` / `fast_mode`, refused until round 5, run in fp32 now:
    test_fp32_training_vilbert_pairs_and_fast_mode_golden.)"""
    from mmf_amd import fp32_path as P
    P._check_head(64, 512); P._check_head(128, 256)
    with pytest.raises(NotImplementedError, match="exceed"):
        P._check_head(64, 513)
    with pytest.raises(NotImplementedError, match="exceed"):
        P._check_head(128, 300)


@pytest.mark.parametrize("name", ["vilbert_pairs", "vilbert_fast"])
def test_fp32_training_vilbert_pairs_and_fast_mode_golden(name):
    """`in_batch_pairs: true` (vilbert.py:678-710: every text against every image, B^2 score rows) and `fast_mode: true` (:712-723: one text against B
    images) on the fp32 kernels — the batch expansion and its backward (the sum over the broadcast index) in fp32 (mmf_expand_batch_f32 /
    mmf_reduce_batch_f32): scores and loss against the reference's fixture, every parameter gradient against the pinned oracle's autograd."""
    from oracle import vilbert_oracle as VO
    from tests.model_utils import build_vilbert
    z, case, cfg, sd, sample = G.load_vilbert_case(name)
    model = build_vilbert(cfg, sd)
    model.eval()
    with mmf_amd.fp32_training():
        if name == "vilbert_pairs":      # B^2 target rows do not pass SampleList's equal-batch check (the reference's neither): the loss is applied here
            out = model(SampleList(sample_to({k: v for k, v in sample.items() if k != "targets"}, "cuda")))
            assert out["scores"].shape[0] == sample["input_ids"].shape[0] ** 2
        else:                            # one text, B images: ViLBERTForClassification.forward directly, as the fixture's generator does
            p = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in VO.prepare_inputs(dict(sample)).items()}
            assert p["input_ids"].shape[0] == 1 and p["image_feature"].shape[0] == 3
            out = model.model(p["input_ids"], p["image_feature"], p["image_location"], p["token_type_ids"], p["attention_mask"], p["image_attention_mask"])
        assert out["scores"].dtype == torch.float32
        loss = mmf_amd.fp32_train.logit_bce(out["scores"], sample["targets"].cuda())
    np.testing.assert_allclose(out["scores"].detach().cpu().numpy(), z["scores"], rtol=TOL_FP32, atol=TOL_FP32)
    assert abs(loss.item() - float(z["loss"])) <= TOL_FP32 * abs(float(z["loss"]))
    loss.backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    O.logit_bce(VO.vilbert_forward(sdr, cfg, dict(sample))["scores"], sample["targets"]).backward()
    errs = _vilbert_grad_check(model, sdr, TOL_FP32)
    assert len(errs) > 50


def test_fp32_gate_kernels_match_float64_autograd():
    """ViLBERT dynamic_attention on fp32 rows (vilbert.py:204-212): masked mean backward, per-sample column gate backward."""
    B, T, H, S, C = 3, 17, 96, 11, 64
    x = _rand(B, T, H, seed=1).cuda(); m = (torch.rand(B, T) > 0.3).float(); m[:, 0] = 1
    g = _rand(B, H, seed=2).cuda()
    dx = torch.full((B * T, H), float("nan"), device="cuda")
    nat.masked_mean_f32_bwd(g, m.cuda(), dx, B, T, H)
    xd = x.double().cpu().requires_grad_(True)
    ((xd * m.double()[..., None]).sum(1) / m.double().sum(1, keepdim=True)).backward(g.double().cpu())
    torch.testing.assert_close(dx.view(B, T, H).cpu().double(), xd.grad, rtol=KERNEL_TOL, atol=KERNEL_TOL)
    ld = 3 * C // 2 * 2 + 8
    raw = _rand(B * S, ld, seed=3); gate = 1.0 + torch.rand(B, C); dy = _rand(B * S, ld, seed=4)
    rd = raw.double().requires_grad_(True); gd = gate.double().requires_grad_(True)
    y = rd.clone()
    y[:, :C] = (rd[:, :C].view(B, S, C) * gd[:, None, :]).reshape(B * S, C)
    y.backward(dy.double())
    yc = y.detach().float().cuda().contiguous(); dyc = dy.cuda().contiguous(); dgate = torch.full((B, C), float("nan"), device="cuda")
    nat.rowgroup_scale_f32_bwd(dyc, yc, ld, gate.cuda(), dgate, B, S, C)
    torch.testing.assert_close(dyc.cpu().double(), rd.grad, rtol=KERNEL_TOL, atol=KERNEL_TOL)
    torch.testing.assert_close(dgate.cpu().double(), gd.grad, rtol=2 * KERNEL_TOL, atol=2 * KERNEL_TOL)


def test_fp32_m4c_kernels_match_float64_autograd():
    """F.normalize backward and the OCR pointer scores' backward on fp32 rows (m4c.py:195, 474-493); the column-slice copy."""
    rows, D, ld = 37, 300, 304
    x = _rand(rows, D, seed=5); g = _rand(rows, ld, seed=6)
    xd = x.double().requires_grad_(True)
    yd = torch.nn.functional.normalize(xd, dim=-1)
    yd.backward(g[:, :D].double())
    y = torch.zeros(rows, ld, device="cuda")
    nat.l2norm_rows_f32(x.cuda(), D, y, ld, rows, D)
    dx = torch.full((rows, D), float("nan"), device="cuda")
    nat.l2norm_rows_f32_bwd(g.cuda(), ld, y, ld, x.cuda(), D, dx, D, rows, D)
    torch.testing.assert_close(dx.cpu().double(), xd.grad, rtol=KERNEL_TOL, atol=KERNEL_TOL)
    B, T, N, HQ, V = 3, 12, 50, 128, 10
    q = _rand(B * T, HQ, seed=7); k = _rand(B * N, HQ, seed=8); ds = _rand(B * T, V + N, seed=9)
    qd, kd = q.double().requires_grad_(True), k.double().requires_grad_(True)
    sc = torch.bmm(qd.view(B, T, HQ), kd.view(B, N, HQ).transpose(1, 2)) / math.sqrt(HQ)
    sc.backward(ds[:, V:].double().reshape(B, T, N))
    dq = torch.full((B * T, HQ), float("nan"), device="cuda"); dk = torch.full((B * N, HQ), float("nan"), device="cuda")
    nat.ptr_scores_f32_bwd(ds.cuda()[:, V:], V + N, q.cuda(), k.cuda(), dq, dk, B, T, N, HQ, 1.0 / math.sqrt(HQ))
    torch.testing.assert_close(dq.cpu().double(), qd.grad, rtol=KERNEL_TOL, atol=KERNEL_TOL)
    torch.testing.assert_close(dk.cpu().double(), kd.grad, rtol=KERNEL_TOL, atol=KERNEL_TOL)
    dst = torch.full((B * T, 12), float("nan"), device="cuda")
    nat.slice_rows_f32(ds.cuda(), V + N, V, dst, 12, B * T)
    assert torch.equal(dst[:, :V].cpu(), ds[:, :V]) and float(dst[:, V:].abs().max()) == 0.0


@pytest.mark.parametrize("B,heads,S,hd", [(2, 2, 70, 64), (2, 12, 228, 64), (2, 4, 100, 128), (2, 2, 300, 64), (1, 2, 160, 128)])
def test_attention_f32_per_query_mask_matches_float64_autograd(B, heads, S, hd):
    """A materialised [B, 1, S, S] additive mask (hf_layers.py:187-190: `attention_scores + attention_mask`, any broadcastable shape) read per
    (query, key) by the fp32 forward and both backward kernels — round 4 built it on the bf16 kernels only."""
    H = heads * hd
    scale = 1.0 / math.sqrt(hd)
    q, k, v, do = _rand(B * S, H, seed=11), _rand(B * S, H, seed=12), _rand(B * S, H, seed=13), _rand(B * S, H, seed=14)
    vis = (torch.rand(B, S, S) > 0.3).float()
    vis[:, torch.arange(S), torch.arange(S)] = 1
    m3 = (1.0 - vis) * -10000.0
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = _ref_attention(qd, kd, vd, m3[:, None].double(), scale, B, heads, S, S, hd)
    ref.backward(do.double())
    qc, kc, vc, doc, mc = q.cuda(), k.cuda(), v.cuda(), do.cuda(), m3.cuda().contiguous()
    out = torch.full((B * S, H), float("nan"), device="cuda"); lse = torch.empty(B, heads, S, device="cuda")
    nat.attention_f32_fwd(qc, kc, vc, H, H, H, mc, out, H, B, heads, S, S, scale, head_dim=hd, lse=lse)
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=KERNEL_TOL, atol=KERNEL_TOL)
    dq, dk, dv = (torch.full_like(t, float("nan")) for t in (qc, kc, vc))
    delta = torch.empty(B, heads, S, device="cuda")
    nat.attention_f32_bwd(qc, kc, vc, H, H, H, mc, out, H, lse, B, heads, S, S, scale, doc, dq, dk, dv, delta, head_dim=hd)
    for name, got, want in (("dq", dq, qd.grad), ("dk", dk, kd.grad), ("dv", dv, vd.grad)):
        torch.testing.assert_close(got.cpu().double(), want, rtol=2 * KERNEL_TOL, atol=2 * KERNEL_TOL, msg=lambda m, n=name: n + ": " + m)
    # one mask per head, [B, heads, S, S] (mmf_attn_desc.mask_head_stride)
    vis4 = (torch.rand(B, heads, S, S) > 0.3).float()
    vis4[:, :, torch.arange(S), torch.arange(S)] = 1
    m4 = (1.0 - vis4) * -10000.0
    qd2, kd2, vd2 = (t.double().requires_grad_(True) for t in (q, k, v))
    ref4 = _ref_attention(qd2, kd2, vd2, m4.double(), scale, B, heads, S, S, hd)
    ref4.backward(do.double())
    m4c = m4.cuda().contiguous()
    nat.attention_f32_fwd(qc, kc, vc, H, H, H, m4c, out, H, B, heads, S, S, scale, head_dim=hd, lse=lse)
    torch.testing.assert_close(out.cpu().double(), ref4.detach(), rtol=KERNEL_TOL, atol=KERNEL_TOL)
    nat.attention_f32_bwd(qc, kc, vc, H, H, H, m4c, out, H, lse, B, heads, S, S, scale, doc, dq, dk, dv, delta, head_dim=hd)
    for name, got, want in (("dq", dq, qd2.grad), ("dk", dk, kd2.grad), ("dv", dv, vd2.grad)):
        torch.testing.assert_close(got.cpu().double(), want, rtol=2 * KERNEL_TOL, atol=2 * KERNEL_TOL, msg=lambda m, n=name: "per-head " + n + ": " + m)
    # the same mask as a key mask (all queries alike) gives the key-mask kernels' result
    key_only = m3[:, :1, :].expand(B, S, S).contiguous().cuda()
    o1 = torch.empty_like(out); o2 = torch.empty_like(out)
    nat.attention_f32_fwd(qc, kc, vc, H, H, H, key_only, o1, H, B, heads, S, S, scale, head_dim=hd)
    nat.attention_f32_fwd(qc, kc, vc, H, H, H, m3[:, 0, :].contiguous().cuda(), o2, H, B, heads, S, S, scale, head_dim=hd)
    assert torch.equal(o1, o2)


def test_fp32_training_image_text_alignment_golden():
    """`image_text_alignment` [B, R, A] (embeddings.py:373-410): the reference's own run (fixture align64), forward + backward in fp32 — the text
    position table collects the aligned regions' gradients."""
    z, case, cfg, sd, sample = G.load_case("align64")
    model = build_visual_bert(cfg, sd)
    model.eval()
    with mmf_amd.fp32_training():
        out = model(SampleList(sample_to(sample, "cuda")))
    np.testing.assert_allclose(out["scores"].detach().cpu().numpy(), z["scores"], rtol=TOL_FP32, atol=TOL_FP32)
    (key, loss), = out["losses"].items()
    assert abs(loss.item() - float(z["loss"])) <= TOL_FP32 * abs(float(z["loss"]))
    loss.backward()
    assert _golden_norm_check(z, dict(model.named_parameters())) > 30
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    O.logit_bce(O.visual_bert_forward(sdo, cfg, sample)["scores"], sample["targets"]).backward()
    key = "bert.embeddings.position_embeddings.weight"
    assert _rel(dict(model.named_parameters())["model." + key].grad, sdo[key].grad) <= TOL_FP32
    with mmf_amd.fp32_inference():      # and the fp32-accurate forward takes the alignment too
        ev = model(SampleList(sample_to(sample, "cuda")))
    np.testing.assert_allclose(ev["scores"].cpu().numpy(), z["scores"], rtol=TOL_FP32, atol=TOL_FP32)


def test_fp32_training_vilbert_dynamic_attention_golden():
    """`dynamic_attention: true` (vilbert.py:174-176, 199-212): text pooling -> dyLinear_q / dyLinear_k -> 1 + sigmoid -> per-sample gates on the
    visual stream's queries and keys, forward + backward on the fp32 kernels: scores / loss against the reference's fixture, every parameter
    gradient (the gate projections included) against the pinned oracle's autograd."""
    from oracle import vilbert_oracle as VO
    from tests.model_utils import build_vilbert
    z, case, cfg, sd, sample = G.load_vilbert_case("vilbert_dyn")
    model = build_vilbert(cfg, sd)
    model.eval()
    with mmf_amd.fp32_training():
        out = model(SampleList(sample_to(sample, "cuda")))
    np.testing.assert_allclose(out["scores"].detach().cpu().numpy(), z["scores"], rtol=TOL_FP32, atol=TOL_FP32)
    (key, loss), = out["losses"].items()
    assert abs(loss.item() - float(z["loss"])) <= TOL_FP32 * abs(float(z["loss"]))
    loss.backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    O.logit_bce(VO.vilbert_forward(sdr, cfg, dict(sample))["scores"], sample["targets"]).backward()
    errs = _vilbert_grad_check(model, sdr, TOL_FP32)
    assert any("dyLinear_q" in k for k in errs) and any("dyLinear_k" in k for k in errs) and len(errs) > 50


@pytest.mark.parametrize("visual_target", [0, 1, 2])
def test_fp32_training_vilbert_masked_region_head_golden(visual_target):
    """ViLBERTForPretraining (vilbert.py:1054-1240): masked LM on the text stream + the masked-region loss on the visual stream — `visual_target` 0 (the
    reference's default: KL against the detector's class distribution), 1 (regression, nn.MSELoss) and 2 (NCE against sampled negatives, the reference
    run's draws replayed) — forward + backward in fp32 against the reference's own runs: both losses, every gradient norm."""
    from tests.model_utils import build_vilbert_pretraining
    z, case, cfg, sd, sample = G.load_vilbert_pretraining_case(visual_target)
    over = dict(num_negative=cfg["num_negative"]) if visual_target == 2 else {}
    model = build_vilbert_pretraining(cfg, sd, visual_target=visual_target, **over)
    model.eval()
    sample = {k: v for k, v in sample.items() if not k.startswith("_")}
    import contextlib
    with (G.recorded_random(z) if visual_target == 2 else contextlib.nullcontext()), mmf_amd.fp32_training():
        out = model(SampleList(sample_to(sample, "cuda")))
    ref = dict(zip((str(k) for k in z["loss_keys"]), z["loss_values"]))
    assert set(out["losses"]) == set(ref)
    for k, v in out["losses"].items():
        assert abs(v.item() - ref[k]) <= TOL_FP32 * abs(ref[k]), (k, v.item(), ref[k])
    sum(v.sum() for v in out["losses"].values()).backward()
    params = dict(model.named_parameters())
    checked = 0
    for gname, norm in zip(z["grad_names"], z["grad_norms"]):
        gname = str(gname)
        p = params[gname]
        if norm == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, gname
            continue
        if gname.endswith(".key.bias") or gname.endswith("key1.bias") or gname.endswith("key2.bias"):
            continue
        assert p.grad is not None and p.grad.dtype == torch.float32, gname
        assert abs(float(p.grad.double().norm()) - norm) <= TOL_FP32 * norm, (gname, float(p.grad.double().norm()), norm)
        full = "grad::" + gname
        if full in z.files and norm > 1e-6:
            assert _rel(p.grad, torch.from_numpy(z[full])) <= TOL_FP32, gname
        checked += 1
    assert checked > 50
    if visual_target == 0:
        with mmf_amd.fp32_inference():
            ev = model(SampleList(sample_to(sample, "cuda")))
        for k, v in ev["losses"].items():
            assert abs(v.item() - ref[k]) <= TOL_FP32 * abs(ref[k]), (k, v.item(), ref[k])


def test_fp32_training_m4c_golden():
    """M4C's stages (m4c.py:185-304) in fp32, forward + backward: row normalisation, OCR feature concat, the padded 3002-wide projection, the
    two-source gather of PrevPredEmbeddings (the classifier weight IS its lookup table: one fp32 gradient = GEMM weight gradient + scattered
    rows), prefix-LM attention, classifier + pointer scores, M4CDecodingBCEWithMaskLoss — against the reference's own run (teacher forcing)."""
    from tests.model_utils import build_m4c
    z, case, cfg, sd, sample = G.load_m4c_case()
    model = build_m4c(cfg, sd)
    model.eval()
    model.training = True          # teacher forcing with every dropout off, as the fixture's generator runs the reference
    with mmf_amd.fp32_training():
        out = model(SampleList(sample_to(sample, "cuda")))
    np.testing.assert_allclose(out["scores"].detach().cpu().numpy(), z["scores"], rtol=TOL_FP32, atol=TOL_FP32)
    (key, loss), = out["losses"].items()
    assert abs(loss.sum().item() - float(z["loss"])) <= TOL_FP32 * abs(float(z["loss"]))
    loss.sum().backward()
    params = dict(model.named_parameters())
    checked = 0
    for gname, norm in zip(z["grad_names"], z["grad_norms"]):
        gname = str(gname)
        p = params[gname]
        assert p.grad is not None and p.grad.dtype == torch.float32, gname
        if gname.endswith(".key.bias") and "ocr_ptr_net" not in gname:      # zero in exact arithmetic: noise vs noise
            continue
        assert abs(float(p.grad.double().norm()) - norm) <= TOL_FP32 * norm, (gname, float(p.grad.double().norm()), norm)
        full = "grad::" + gname
        if full in z.files:
            assert _rel(p.grad, torch.from_numpy(z[full])) <= TOL_FP32, gname
        checked += 1
    assert checked > 40 and "classifier.module.weight" in params
    assert float(params["text_bert.embeddings.word_embeddings.weight"].grad[0].abs().max()) == 0.0     # [PAD]


def test_fp32_training_visual_bert_pretraining_golden():
    """`training_head_type: pretraining` (masked LM over the joint sequence, decoder tied to the word embeddings): loss and logits against the
    reference's fixture, every gradient norm — the tied table collects the gather's AND the decoder's gradient."""
    from tests.model_utils import build_visual_bert_pretraining
    z, case, cfg, sd, sample = G.load_pretraining_case()
    model = build_visual_bert_pretraining(cfg, sd)
    model.eval()
    with mmf_amd.fp32_training():
        out = model(SampleList(sample_to(sample, "cuda")))
    (key, loss), = out["losses"].items()
    assert abs(loss.item() - float(z["loss"])) <= TOL_FP32 * abs(float(z["loss"]))
    if "logits" in z.files and "logits" in out:
        np.testing.assert_allclose(out["logits"].detach().cpu().numpy().reshape(z["logits"].shape), z["logits"], rtol=TOL_FP32, atol=TOL_FP32)
    loss.backward()
    assert _golden_norm_check(z, dict(model.named_parameters())) > 30


def _vilbert_grad_check(model, sdr, tol):
    params = dict(model.named_parameters())
    errs = {}
    for k, v in sdr.items():
        p = params["model." + k]
        if v.grad is None or float(v.grad.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None and p.grad.dtype == torch.float32, k
        if k.endswith(".key.bias") or k.endswith("key1.bias") or k.endswith("key2.bias"):      # zero in exact arithmetic: noise vs noise
            continue
        errs[k] = _rel(p.grad, v.grad)
    bad = {k: e for k, e in errs.items() if e > tol}
    assert not bad, bad
    return errs


@pytest.mark.parametrize("name", ["vilbert_small", "vilbert_nlvr2"])
def test_fp32_training_vilbert_golden(name):
    """ViLBERT (two streams, co-attention with Sq != Sk, ReLU poolers, `mul` fusion, nlvr2 pairing) forward + backward on the fp32 kernels:
    loss and scores against the reference's fixture, every parameter gradient against the pinned oracle's autograd."""
    from oracle import vilbert_oracle as VO
    from tests.model_utils import build_vilbert
    z, case, cfg, sd, sample = G.load_vilbert_case(name)
    nl = name == "vilbert_nlvr2"
    model = build_vilbert(cfg, sd, **(dict(training_head_type="nlvr2", losses=[dict(type="cross_entropy")]) if nl else {}))
    model.eval()
    with mmf_amd.fp32_training():
        out = model(SampleList(sample_to(sample, "cuda")))
    np.testing.assert_allclose(out["scores"].detach().cpu().numpy(), z["scores"], rtol=TOL_FP32, atol=TOL_FP32)
    (key, loss), = out["losses"].items()
    assert abs(loss.item() - float(z["loss"])) <= TOL_FP32 * abs(float(z["loss"]))
    loss.backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = VO.vilbert_forward(sdr, cfg, dict(sample))
    ref_loss = torch.nn.functional.cross_entropy(ref["scores"], sample["targets"]) if nl else O.logit_bce(ref["scores"], sample["targets"])
    ref_loss.backward()
    errs = _vilbert_grad_check(model, sdr, TOL_FP32)
    assert len(errs) > 50


def test_fp32_training_vilbert_real_widths_with_dropout_runs():
    """BASELINE configs[3]'s widths (768 / 12 + 1024 / 8 + co-attention 1024 / 8: head_dim 128), train mode with dropout: gradients are finite,
    reproducible under the same seed, and a fused AdamW step lowers the loss on the fixed batch."""
    from oracle import vilbert_oracle as VO
    from mmf_amd.modules.optimizers import AdamW
    from tests.model_utils import build_vilbert
    cfg = dict(VO.DEFAULT_CONFIG)
    cfg.update(num_hidden_layers=3, v_num_hidden_layers=2, v_biattention_id=[0, 1], t_biattention_id=[1, 2], vocab_size=2000,
               max_position_embeddings=128, initializer_range=0.02)
    g = torch.Generator().manual_seed(5)
    sd = {k: (1.0 + 0.05 * torch.randn(shp, generator=g)) if "LayerNorm" in k and k.endswith(".weight") else 0.02 * torch.randn(shp, generator=g)
          for k, shp in VO.parameter_shapes(cfg).items()}
    B, T, R = 4, 128, 100
    ids = torch.randint(1, cfg["vocab_size"], (B, T), generator=g)
    targets = torch.zeros(B, cfg["num_labels"]); targets[0, 5] = 1.0; targets[1, 17] = 0.6
    sample = {"input_ids": ids, "input_mask": torch.ones(B, T, dtype=torch.long), "segment_ids": torch.zeros(B, T, dtype=torch.long),
              "image_feature_0": torch.randn(B, R, cfg["v_feature_size"], generator=g),
              "image_info_0": {"max_features": torch.tensor([100, 73, 100, 12]), "bbox": torch.rand(B, R, 5, generator=g)},
              "targets": targets, "dataset_name": "vqa2", "dataset_type": "train"}
    torch.manual_seed(7)
    model = build_vilbert(cfg, sd)
    model.train()
    opt = AdamW(model.parameters(), lr=1e-3)
    batch = SampleList(sample_to(sample, "cuda"))
    losses = []
    for _ in range(4):
        opt.zero_grad()
        with mmf_amd.fp32_training():
            out = model(batch)
        loss = list(out["losses"].values())[0]
        loss.backward()
        assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in model.parameters())
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0], losses


def _golden_norm_check(z, params, alias=None):
    checked = 0
    for gname, norm in zip(z["grad_names"], z["grad_norms"]):
        gname = str(gname)
        p = params[(alias or {}).get(gname, gname)]
        if norm == 0.0:      # (MMBT decoder mode: query / key gradients are sums of equal terms with opposite signs — exactly 0 in the reference)
            assert p.grad is None or float(p.grad.abs().max()) <= 1e-7, gname
            continue
        if gname.endswith("self.key.bias"):
            continue
        assert p.grad is not None and p.grad.dtype == torch.float32, gname
        assert abs(float(p.grad.double().norm()) - norm) <= TOL_FP32 * norm, (gname, float(p.grad.double().norm()), norm)
        full = "grad::" + gname
        if full in z.files:
            assert _rel(p.grad, torch.from_numpy(z[full])) <= TOL_FP32, gname
        checked += 1
    return checked


@pytest.mark.parametrize("name", ["mmbt_small64", "mmbt_decoder64"])
def test_fp32_training_mmbt_golden(name):
    """BASELINE.json configs[0] (MMBT): modal block (start token, projected features with their position rows and the modal type row, end
    token) + text through the fp32 encoder, forward + backward, against the reference's fixture; `mmbt_decoder64`: decoder mode (mmbt.py:244-272,
    the causal per-query mask read by the fp32 attention kernels)."""
    from oracle.mmbt_oracle import SHARED
    from tests.model_utils import build_mmbt
    z, case, cfg, sd, sample = G.load_mmbt_case(name)
    model = build_mmbt(cfg, sd, SHARED)
    model.eval()
    with mmf_amd.fp32_training():
        out = model(SampleList(sample_to(sample, "cuda")))
    np.testing.assert_allclose(out["scores"].detach().cpu().numpy(), z["scores"], rtol=TOL_FP32, atol=TOL_FP32)
    (key, loss), = out["losses"].items()
    assert abs(loss.item() - float(z["loss"])) <= TOL_FP32 * abs(float(z["loss"]))
    loss.backward()
    assert _golden_norm_check(z, dict(model.named_parameters())) > 30


def test_fp32_training_mmft_golden():
    """MMF Transformer (text block + Linear -> LayerNorm image tokens with position / type rows, concat, encoder, MLP head), forward +
    backward on the fp32 kernels, against the reference's fixture; the [PAD] row of the word table gets no gradient."""
    from oracle import mmft_oracle
    from tests.model_utils import build_mmft
    z, case, cfg, sd, sample = G.load_mmft_case()
    model = build_mmft(cfg, sd, mmft_oracle.shared(cfg))
    model.eval()
    with mmf_amd.fp32_training():
        out = model(SampleList(sample_to(sample, "cuda")))
    np.testing.assert_allclose(out["scores"].detach().cpu().numpy(), z["scores"], rtol=TOL_FP32, atol=TOL_FP32)
    (key, loss), = out["losses"].items()
    assert abs(loss.item() - float(z["loss"])) <= TOL_FP32 * abs(float(z["loss"]))
    loss.backward()
    params = dict(model.named_parameters())
    assert _golden_norm_check(z, params, mmft_oracle.shared(cfg)) > 30
    assert float(params["backend.transformer.embeddings.word_embeddings.weight"].grad[0].abs().max()) == 0.0


def test_fp32_training_uniter_golden():
    """UNITER (feature + mask-embedding rows, 7-d box geometry Linear, three LayerNorms, text block, concat, encoder, MLP head): forward +
    backward on the fp32 kernels against the reference's fixture."""
    from tests.model_utils import build_uniter
    z, case, cfg, sd, sample = G.load_uniter_case()
    model = build_uniter(cfg, sd)
    model.eval()
    with mmf_amd.fp32_training():
        out = model(SampleList(sample_to(sample, "cuda")))
    np.testing.assert_allclose(out["scores"].detach().cpu().numpy(), z["scores"], rtol=TOL_FP32, atol=TOL_FP32)
    (key, loss), = out["losses"].items()
    assert abs(loss.sum().item() - float(z["loss"])) <= TOL_FP32 * abs(float(z["loss"]))
    loss.sum().backward()
    assert _golden_norm_check(z, dict(model.named_parameters())) > 30
