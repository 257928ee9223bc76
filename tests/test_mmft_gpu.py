"""MMF Transformer on the HIP path (GPU): the two kernels it adds (per-modality embedding sum, concat/split copies),
the fp32-feature Linear, and the registered `mmft` model against the fixture recorded from the real reference
(tests/golden/mmft_small64.npz) and the CPU oracle.  Tolerance: BASELINE.json north_star, 5e-2 for the bf16 path."""
import numpy as np
import pytest
import torch

from oracle import mmft_oracle as O
from tests.golden_utils import load_mmft_case
from tests.model_utils import build_mmft, sample_to
from tests.test_kernels_gpu import close, nat, rnd, DEV
from mmf_amd.common.sample import SampleList

pytestmark = pytest.mark.gpu
TOL = 5e-2


def rel_err(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def test_rows_add_embed_and_its_backward():
    import mmf_amd.functional as Fn
    B, L, H = 3, 7, 256
    x = rnd(B, L, H).requires_grad_(True)
    seg = torch.randint(0, 3, (B, L), device=DEV)
    pos = rnd(16, H, dtype=torch.float32).requires_grad_(True); typ = rnd(3, H, dtype=torch.float32).requires_grad_(True)
    y = Fn.AddPosTypeFn.apply(x, seg, pos, typ)
    ref = x.detach().float() + pos.detach()[:L][None] + typ.detach()[seg]
    close(y, ref, 1e-2, 1e-2, "add pos/type")
    g = rnd(B, L, H, seed=5)
    y.backward(g)
    assert torch.equal(x.grad, g)
    close(pos.grad[:L], g.float().sum(0), 1e-4, 1e-3, "dpos")
    assert float(pos.grad[L:].abs().max()) == 0
    close(typ.grad, torch.zeros(3, H, device=DEV).index_add_(0, seg.view(-1), g.float().view(-1, H)), 1e-4, 1e-3, "dtype")
    # no segment ids / no positions
    y2 = Fn.AddPosTypeFn.apply(x.detach(), None, pos.detach(), None)
    close(y2, x.detach().float() + pos.detach()[:L][None], 1e-2, 1e-2, "add pos only")


def test_concat_rows_forward_and_split_backward():
    import mmf_amd.functional as Fn
    B, H = 4, 128
    a = rnd(B, 5, H).requires_grad_(True); b = rnd(B, 1, H).requires_grad_(True); c = rnd(B, 9, H).requires_grad_(True)
    out = Fn.ConcatRowsFn.apply(a, b, c)
    assert torch.equal(out, torch.cat([a, b, c], dim=1).detach())
    g = rnd(B, 15, H, seed=3)
    out.backward(g)
    assert torch.equal(a.grad, g[:, :5]) and torch.equal(b.grad, g[:, 5:6]) and torch.equal(c.grad, g[:, 6:])


def test_linear_takes_raw_fp32_features():
    import mmf_amd.functional as Fn
    M, K, N = 300, 72, 128
    x = rnd(M, K, dtype=torch.float32)
    w = rnd(N, K, dtype=torch.float32, scale=0.1).requires_grad_(True); bias = rnd(N, dtype=torch.float32).requires_grad_(True)
    y = Fn.linear(x, w, bias)
    ref = x.bfloat16().float() @ w.detach().bfloat16().float().t() + bias.detach()
    close(y, ref, 1e-2, 2e-2, "fp32-input linear")
    g = rnd(M, N, seed=9)
    y.backward(g)
    close(w.grad, g.float().t() @ x.bfloat16().float(), 1e-2, 5e-2, "wgrad with fp32 activations")
    close(bias.grad, g.float().sum(0), 1e-3, 1e-2, "bias grad")


def test_mmft_golden_forward_loss_and_gradients():
    z, case, cfg, sd, sample = load_mmft_case()
    model = build_mmft(cfg, sd, O.shared(cfg))
    model.eval()
    seq = {}
    hook = model.backend.register_forward_hook(lambda m, i, o: seq.update(seq=o[0]))
    out = model(SampleList(sample_to(sample, "cuda")))
    hook.remove()
    np.testing.assert_allclose(out["scores"].detach().float().cpu().numpy(), z["scores"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(seq["seq"].detach().float().cpu().numpy(), z["sequence_output"], rtol=TOL, atol=TOL)
    (key, loss), = out["losses"].items()
    assert key == "train/hateful_memes/cross_entropy"
    assert abs(loss.item() - float(z["loss"])) <= TOL * abs(float(z["loss"]))
    loss.sum().backward()
    params = dict(model.named_parameters())
    alias = O.shared(cfg)
    worst = {}
    for gname, norm in zip(z["grad_names"], z["grad_norms"]):
        gname = str(gname)
        p = params[alias.get(gname, gname)]
        if norm == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, gname
            continue
        assert p.grad is not None, gname
        gn = float(p.grad.double().norm())
        if gname.endswith("self.key.bias"):
            qn = float(params[gname.replace("key.bias", "query.bias")].grad.double().norm())
            assert gn <= TOL * qn + 1e-6, (gname, gn, qn)
            continue
        worst[gname] = abs(gn - norm) / norm
        full = "grad::" + gname
        if full in z.files:
            assert rel_err(p.grad, torch.from_numpy(z[full])) <= TOL, gname
    bad = {k: round(v, 4) for k, v in worst.items() if v > TOL}
    assert not bad, bad
    # padding_idx: the [PAD] row of the word table gets no gradient
    assert float(params["backend.transformer.embeddings.word_embeddings.weight"].grad[0].abs().max()) == 0.0


def test_mmft_all_gradients_match_oracle_three_modalities():
    """Three modalities (text + two feature streams, one without segment ids), every parameter's full gradient
    against the pinned CPU oracle."""
    z, case, cfg, sd, sample = load_mmft_case()
    H = cfg["hidden_size"]
    cfg = dict(cfg)
    cfg["modalities"] = [dict(m) for m in cfg["modalities"]] + [
        dict(type="audio", key="audio", embedding_dim=40, position_dim=cfg["max_position_embeddings"], layer_norm_eps=1e-12,
             hidden_dropout_prob=0.1)]
    g = torch.Generator().manual_seed(77)
    sd = dict(sd)
    for k, shp in O.parameter_shapes(cfg).items():
        if k not in sd or tuple(sd[k].shape) != tuple(shp):
            sd[k] = (1.0 + 0.05 * torch.randn(shp, generator=g)) if k.endswith("1.weight") or "layer_norms" in k and k.endswith("weight") \
                else 0.05 * torch.randn(shp, generator=g)
    sample = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in sample.items()}
    sample["audio"] = torch.randn(sample["input_ids"].shape[0], 5, 40, generator=g)
    model = build_mmft(cfg, sd, O.shared(cfg))
    model.eval()
    out = model(SampleList(sample_to(sample, "cuda")))
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.mmft_forward(sdr, cfg, dict(sample))
    assert float((out["scores"].detach().float().cpu() - ref["scores"].detach()).abs().max()) <= TOL
    (key, loss), = out["losses"].items()
    ref_loss = torch.nn.functional.cross_entropy(ref["scores"], sample["targets"])
    assert abs(loss.item() - ref_loss.item()) <= TOL * abs(ref_loss.item())
    loss.sum().backward(); ref_loss.backward()
    params = dict(model.named_parameters())
    errs = {}
    for k, v in sdr.items():
        p = params[k]
        if v.grad is None or float(v.grad.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        if k.endswith("self.key.bias"):
            continue
        errs[k] = rel_err(p.grad, v.grad)
    bad = {k: round(e, 4) for k, e in errs.items() if e > TOL}
    assert not bad, bad


def test_mmft_bert_large_widths_match_oracle():
    """BASELINE.json configs[3] shape class (MMFT / UNITER at BERT-large widths: H = 1024, 16 heads, I = 4096, joint sequence
    128 tokens + 100 regions x 2048) with two layers: every parameter's full gradient against the pinned CPU oracle."""
    H = 1024
    cfg = dict(O.DEFAULT_CONFIG)
    cfg.update(hidden_size=H, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096, vocab_size=2000,
               max_position_embeddings=128, num_labels=3, initializer_range=0.02)
    cfg["modalities"] = [dict(type="text", key="text", position_dim=128, segment_id=0, embedding_dim=H, layer_norm_eps=1e-12, hidden_dropout_prob=0.1),
                         dict(type="image", key="image", embedding_dim=2048, position_dim=128, segment_id=1, layer_norm_eps=1e-12,
                              hidden_dropout_prob=0.1)]
    g = torch.Generator().manual_seed(11)
    sd = {}
    for k, shp in O.parameter_shapes(cfg).items():
        ln = k.endswith("LayerNorm.weight") or (("layer_norms" in k or k.endswith(".1.weight")) and len(shp) == 1 and k.endswith("weight"))
        sd[k] = (1.0 + 0.05 * torch.randn(shp, generator=g)) if ln else 0.02 * torch.randn(shp, generator=g)
    B, T, R = 2, 128, 100
    ids = torch.randint(1, cfg["vocab_size"], (B, T), generator=g)
    mask = torch.ones(B, T, dtype=torch.long); mask[1, 77:] = 0; ids[mask == 0] = 0
    image_mask = torch.ones(B, R, dtype=torch.long); image_mask[0, 90:] = 0
    sample = {"input_ids": ids, "input_mask": mask, "segment_ids": torch.zeros(B, T, dtype=torch.long),
              "image": torch.randn(B, R, 2048, generator=g), "image_mask": image_mask, "targets": torch.tensor([2, 0]),
              "dataset_name": "hateful_memes", "dataset_type": "train"}
    model = build_mmft(cfg, sd, O.shared(cfg))
    model.eval()
    out = model(SampleList(sample_to(sample, "cuda")))
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.mmft_forward(sdr, cfg, dict(sample))
    assert rel_err(out["scores"], ref["scores"]) <= TOL
    (key, loss), = out["losses"].items()
    ref_loss = torch.nn.functional.cross_entropy(ref["scores"], sample["targets"])
    assert abs(loss.item() - ref_loss.item()) <= TOL * abs(ref_loss.item())
    loss.sum().backward(); ref_loss.backward()
    params = dict(model.named_parameters())
    errs = {}
    for k, v in sdr.items():
        p = params[k]
        if v.grad is None or float(v.grad.abs().max()) == 0.0:
            continue
        assert p.grad is not None, k
        if k.endswith("self.key.bias"):
            continue
        errs[k] = rel_err(p.grad, v.grad)
    bad = {k: round(e, 4) for k, e in errs.items() if e > TOL}
    assert not bad, bad


def test_mmft_training_mode_is_seed_reproducible():
    z, case, cfg, sd, sample = load_mmft_case()
    model = build_mmft(cfg, sd, O.shared(cfg))
    model.train()
    batch = sample_to(sample, "cuda")
    torch.manual_seed(3)
    a = model(SampleList(dict(batch)))["scores"].float().clone()
    torch.manual_seed(3)
    b = model(SampleList(dict(batch)))["scores"].float().clone()
    torch.manual_seed(4)
    c = model(SampleList(dict(batch)))["scores"].float().clone()
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_mlm_and_itm_heads_match_the_reference_heads():
    """`mlm` / `itm` transformer heads (mmf/models/transformers/heads/{mlm,itm}.py) stand-alone against the fixture recorded from the
    reference heads: logits of the masked rows only, both losses, parameter gradients, the tied table's gradient and the gradient
    handed back to the encoder (zero on the unmasked rows for MLM, row 0 only for ITM); nothing masked -> loss 0 with a warning."""
    import warnings
    from mmf_amd.models.transformers.heads.itm import ITM
    from mmf_amd.models.transformers.heads.mlm import MLM
    from tests.golden_utils import load_transformer_heads_case
    z, case, sds, inp = load_transformer_heads_case()
    V, H = case["vocab_size"], case["hidden_size"]
    table = torch.nn.Embedding(V, H)
    table.load_state_dict(sds["table"])
    mlm = MLM(dict(type="mlm", vocab_size=V, hidden_size=H))
    mlm.tie_weights(table)
    full = dict(sds["mlm"])
    full["cls.predictions.decoder.weight"] = sds["table"]["weight"]
    full["cls.predictions.decoder.bias"] = full["cls.predictions.bias"]
    mlm.load_state_dict(full, strict=True)
    itm = ITM(dict(type="itm", hidden_size=H))
    itm.load_state_dict(sds["itm"], strict=True)
    table, mlm, itm = table.cuda(), mlm.cuda().eval(), itm.cuda().eval()
    mlm.tie_weights(table)
    seq = inp["sequence_output"].cuda().requires_grad_(True)
    proc = {"mlm_labels": {"combined_labels": inp["labels"].cuda()}, "itm_labels": {"is_correct": inp["is_correct"].cuda()}}
    a = mlm(seq, processed_sample_list=proc)
    b = itm(seq, processed_sample_list=proc)
    assert a["logits"].shape == z["mlm_logits"].shape
    np.testing.assert_allclose(a["logits"].detach().cpu().numpy(), z["mlm_logits"], rtol=TOL, atol=TOL)
    assert abs(a["losses"]["masked_lm_loss"].item() - float(z["mlm_loss"])) <= TOL * float(z["mlm_loss"])
    assert abs(b["losses"]["itm_loss"].item() - float(z["itm_loss"])) <= TOL * float(z["itm_loss"])
    (a["losses"]["masked_lm_loss"] + b["losses"]["itm_loss"]).backward()
    assert rel_err(seq.grad, torch.from_numpy(z["grad_sequence_output"])) <= TOL
    assert rel_err(table.weight.grad, torch.from_numpy(z["grad::table.weight"])) <= TOL
    for tag, mod in (("mlm", mlm), ("itm", itm)):
        for k, p in mod.named_parameters():
            key = "grad::%s.%s" % (tag, k)
            if key in z.files:
                assert rel_err(p.grad, torch.from_numpy(z[key])) <= TOL, key
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        blank = mlm(seq, processed_sample_list={"mlm_labels": {"combined_labels": torch.full_like(inp["labels"], -1).cuda()}})
    assert blank["losses"]["masked_lm_loss"].item() == 0.0 and blank["logits"].shape == (0, V)
    assert any("NaN detected in masked_lm_loss" in str(x.message) for x in w)
    # the zero stays attached to the graph as the reference's nan_to_num(cross_entropy over zero rows) does: backward works and the
    # encoder output and every head parameter get an all-zero gradient (not None)
    seq.grad = None
    for p in mlm.parameters():
        p.grad = None
    blank["losses"]["masked_lm_loss"].backward()
    assert seq.grad is not None and float(seq.grad.abs().max()) == 0.0
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in mlm.parameters())


def test_mmft_pretraining_step_with_mlm_and_itm_heads():
    """MMF Transformer with `heads: [mlm, itm]` (decoder tied to the text token embedding): a few fused-AdamW steps on a fixed batch,
    the summed loss goes down and the tied table receives the gather + decoder gradient."""
    from mmf_amd.common.registry import registry
    from mmf_amd.utils.configuration import Config
    from tests.model_utils import mmft_model_config
    from mmf_amd.utils.build import build_model
    z, case, cfg, sd, sample = load_mmft_case()
    model = build_model(mmft_model_config(cfg, heads=[
        dict(type="mlm", vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"]), dict(type="itm", hidden_size=cfg["hidden_size"])],
        losses=[])).cuda()
    model.train()
    g = torch.Generator().manual_seed(11)
    B, T = sample["input_ids"].shape
    lm = torch.where(torch.rand(B, T, generator=g) < 0.3, sample["input_ids"], torch.full((B, T), -1))
    lm[:, 1] = sample["input_ids"][:, 1]
    s = dict(sample, lm_label_ids=lm, is_correct=torch.tensor([1, 0, 1])[:B])
    s.pop("targets", None)
    batch = SampleList(sample_to(s, "cuda"))
    full = Config(model="mmft", optimizer=dict(params=dict(lr=1e-3)), model_config=dict(mmft=model.config))
    opt = registry.get_optimizer_class("adam_w")(model.get_optimizer_parameters(full), lr=1e-3, eps=1e-8)
    totals = []
    out = model(batch)
    # mmf_transformer.py:426-431 `update`s ONE dict per head, so the last head's `losses` survives (here as there): the two heads are
    # driven directly below to train on both losses
    assert sorted(out["losses"]) == ["itm_loss"] and out["logits"].shape[1] == cfg["vocab_size"]
    del out
    for _ in range(8):
        processed = model.preprocess_sample(batch)
        masks = [processed["masks"][m] for m in model.modality_keys]
        seq, layers = model.backend(processed["input_ids"], processed["position_ids"], processed["segment_ids"], masks)
        la = model.heads[0](seq, layers, processed)["losses"]["masked_lm_loss"]
        lb = model.heads[1](seq, layers, processed)["losses"]["itm_loss"]
        loss = la + lb
        opt.zero_grad()
        loss.backward()
        opt.step()
        totals.append(loss.item())
    assert all(np.isfinite(totals)) and totals[-1] < totals[0] - 0.5, totals
    w = model.backend.embeddings.token_embeddings[0].weight
    assert model.heads[0].cls.predictions.decoder.weight is w and w.grad is not None and bool(torch.isfinite(w.grad).all())


@pytest.mark.parametrize("use_kl", [True, False])
def test_mrc_head_matches_the_reference_head(use_kl):
    """`mrc` (masked region classification, mmf/models/transformers/heads/mrc.py) against the reference head's own run: loss, parameter
    gradients and the gradient handed back to the encoder (zero outside the masked regions), both loss variants."""
    from mmf_amd.models.transformers.heads.mrc import MRC
    from tests.golden_utils import load_transformer_heads_case
    z, case, sds, inp = load_transformer_heads_case()
    tag = "kl" if use_kl else "ce"
    head = MRC(hidden_size=case["hidden_size"], label_dim=inp["region_class"].shape[1], use_kl=use_kl)
    assert sorted(head.state_dict().keys()) == sorted(sds["mrc"].keys())
    head.load_state_dict(sds["mrc"], strict=True)
    head = head.cuda().eval()
    seq = inp["sequence_output"].cuda().requires_grad_(True)
    out = head(seq, {"region_class": inp["region_class"].cuda(), "image_region_mask": inp["region_mask"].cuda()})
    loss = out["losses"]["mrc_loss"]
    assert loss.shape == torch.Size([]) and abs(loss.item() - float(z["mrc_%s_loss" % tag])) <= TOL * float(z["mrc_%s_loss" % tag])
    loss.backward()
    ref_g = torch.from_numpy(z["mrc_%s_grad_sequence_output" % tag])
    assert rel_err(seq.grad, ref_g) <= TOL
    assert float(seq.grad.float().cpu()[~inp["region_mask"]].abs().max()) == 0.0
    for k, p in head.named_parameters():
        assert rel_err(p.grad, torch.from_numpy(z["grad::mrc_%s.%s" % (tag, k)])) <= TOL, k


def test_reference_head_unit_tests_ported():
    """tests/models/transformers/test_heads.py of the reference (MLM :21-62, MLP :65-81, ITM :84-103, multilayer MLP :106-128, MRC :204-236):
    output keys and shapes of each head on the shapes those tests use."""
    import warnings
    from mmf_amd.models.transformers.heads.itm import ITM
    from mmf_amd.models.transformers.heads.mlm import MLM
    from mmf_amd.models.transformers.heads.mlp import MLP
    from mmf_amd.models.transformers.heads.mrc import MRC
    x = torch.rand(1, 64, 768, device="cuda")
    module = MLM(dict(type="mlm", freeze=False, vocab_size=1000, hidden_size=768)).cuda()
    out = module(x, [x, x], {"mlm_labels": {"combined_labels": torch.ones(1, 64, dtype=torch.long, device="cuda")}})
    assert "logits" in out and "masked_lm_loss" in out["losses"] and out["logits"].shape == torch.Size([64, 1000])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = module(x, [x, x], {"mlm_labels": {"combined_labels": torch.full((1, 64), module.config.ignore_index, dtype=torch.long,
                                                                                device="cuda")}})
    assert not torch.isnan(out["losses"]["masked_lm_loss"]) and out["losses"]["masked_lm_loss"] == 0.0
    ones = torch.ones(1, 64, 768, device="cuda")
    out = MLP(dict(type="mlp", num_labels=2, hidden_size=768)).cuda()(ones, [ones, ones], {})
    assert "scores" in out and out["scores"].shape == torch.Size([1, 2])
    out = MLP(dict(type="mlp", num_labels=2, hidden_size=768, num_layers=2, in_dim=768, pooler_name="bert_pooler")).cuda()(ones, [ones, ones], {})
    assert out["scores"].shape == torch.Size([1, 2])
    out = ITM(dict(type="itm", hidden_size=768)).cuda()(ones, [ones, ones], {"itm_labels": {"is_correct": torch.tensor(False, dtype=torch.long,
                                                                                                                        device="cuda")}})
    assert "itm_loss" in out["losses"] and out["losses"]["itm_loss"].shape == torch.Size([])
    bs, num_feat, label_dim = 8, 64, 100
    seq = torch.ones(bs, num_feat, 768, device="cuda")
    proc = {"region_class": torch.rand(bs, num_feat, label_dim, device="cuda").view(-1, label_dim),
            "image_region_mask": torch.ones(bs, num_feat, device="cuda").bool()}
    for kw in (dict(), dict(use_kl=False)):
        out = MRC(hidden_size=768, label_dim=label_dim, **kw).cuda()(seq, proc)
        assert "mrc_loss" in out["losses"] and out["losses"]["mrc_loss"].shape == torch.Size([])


def test_mrfr_and_wra_heads_match_the_reference_fixture():
    """The two UNITER pretraining heads built in round 3, stand-alone, against the reference's own heads (tests/golden/make_golden.py,
    `transformer_heads`): MRFR — compaction, Linear-GELU-LayerNorm, the TIED image-embedding weight applied transposed, MSE — loss, the
    gradient of the sequence output, of every head parameter and of the tied weight; WRA — optimal-transport distance with ragged text /
    region lengths (padding masks) and mixed labels — loss and the gradient of the sequence output."""
    import numpy as np
    from mmf_amd.models.transformers.heads.mrfr import MRFR
    from mmf_amd.models.transformers.heads.wra import WRA
    from tests.golden_utils import load_transformer_heads_extra
    z, case, sds, inp = load_transformer_heads_extra()
    seq0 = inp["sequence_output"].cuda()
    H = seq0.shape[-1]
    img_w = torch.nn.Parameter(sds["img"]["weight"].clone().cuda())
    head = MRFR(img_w, hidden_size=H, img_dim=img_w.shape[1]).cuda().eval()
    missing, unexpected = head.load_state_dict({k: v.cuda() for k, v in sds["mrfr"].items()}, strict=False)
    assert not unexpected and missing == ["linear_proj_weight"], (missing, unexpected)
    seq = seq0.clone().requires_grad_(True)
    loss = head(seq, {"mrfr_region_target": inp["mrfr_target"].cuda(), "mrfr_region_mask": inp["region_mask"].cuda()})["losses"]["mrfr_loss"]
    assert abs(loss.item() - float(z["mrfr_loss"])) <= 5e-2 * float(z["mrfr_loss"]), (loss.item(), float(z["mrfr_loss"]))
    loss.backward()

    def rel(a, b):
        b = torch.from_numpy(np.asarray(b)).double()
        return float((a.detach().double().cpu() - b).norm() / (b.norm() + 1e-30))
    assert rel(seq.grad, z["mrfr_grad_sequence_output"]) <= 5e-2
    assert rel(img_w.grad, z["grad::img.weight"]) <= 5e-2
    for k, p in head.named_parameters():
        if k != "linear_proj_weight":
            assert rel(p.grad, z["grad::mrfr." + k]) <= 5e-2, k
    # WRA
    tl, il = inp["txt_pad"].shape[1], inp["img_pad"].shape[1]
    wra = WRA().cuda().eval()
    seq = seq0.clone().requires_grad_(True)
    proc = {"wra_info": {"txt_pad": inp["txt_pad"].cuda(), "img_pad": inp["img_pad"].cuda()}, "is_correct": inp["is_correct"].cuda(),
            "input_ids": torch.zeros(seq0.shape[0], tl, dtype=torch.long, device="cuda"), "image_feat": torch.zeros(seq0.shape[0], il, 4, device="cuda")}
    lw = wra(seq, proc)["losses"]["wra_loss"]
    assert abs(lw.item() - float(z["wra_loss"])) <= 5e-2 * abs(float(z["wra_loss"])), (lw.item(), float(z["wra_loss"]))
    lw.backward()
    assert rel(seq.grad, z["wra_grad_sequence_output"]) <= 5e-2
    # the OT distance itself, per sample, against the oracle on the bf16-rounded sequence (tight: same arithmetic, fp32)
    from oracle import mmft_oracle as HO
    from mmf_amd import functional as Fn
    sq = seq0.bfloat16().float().cpu()
    want = HO.optimal_transport_dist(sq[:, :tl], sq[:, tl:tl + il], inp["txt_pad"].bool(), inp["img_pad"].bool())
    _, dist = Fn.WordRegionAlignmentFn.apply(seq0, tl, il, inp["txt_pad"].cuda(), inp["img_pad"].cuda(), inp["is_correct"].cuda())
    np.testing.assert_allclose(dist.cpu().numpy(), want.numpy(), rtol=2e-3, atol=1e-5)


def test_wra_at_the_uniter_shape_matches_the_oracle():
    """128 tokens + 100 regions, H = 768, B = 8, ragged lengths: distances and the gradient against the oracle (fp32 torch on the host)."""
    from mmf_amd import functional as Fn
    from oracle import mmft_oracle as HO
    g = torch.Generator().manual_seed(5)
    B, M, N, H = 8, 128, 100, 768
    seq = (torch.randn(B, M + N, H, generator=g) * 0.7 + 0.2 * torch.randn(1, 1, H, generator=g)).bfloat16()
    txt_pad = torch.zeros(B, M, dtype=torch.bool); img_pad = torch.zeros(B, N, dtype=torch.bool)
    for b in range(B):
        txt_pad[b, 16 + 14 * b:] = True
        img_pad[b, 100 - 9 * b:] = True
    labels = torch.tensor([1, 0, 1, 1, 0, 0, 1, 0])
    sq = seq.float().clone().requires_grad_(True)
    want = HO.wra_head(sq, M, N, txt_pad, img_pad, labels)
    want["losses"]["wra_loss"].backward()
    x = seq.cuda().requires_grad_(True)
    loss, dist = Fn.WordRegionAlignmentFn.apply(x, M, N, txt_pad.cuda(), img_pad.cuda(), labels.cuda())
    loss.backward()
    assert float((dist.cpu() - want["ot_dist"].detach()).abs().max()) <= 2e-3 * float(want["ot_dist"].abs().max())
    assert abs(loss.item() - want["losses"]["wra_loss"].item()) <= 2e-3 * abs(want["losses"]["wra_loss"].item()) + 1e-5
    gr = sq.grad
    assert float((x.grad.float().cpu() - gr).norm() / gr.norm()) <= 2e-2
