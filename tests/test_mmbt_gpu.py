"""MMBT on the HIP path (GPU): the kernels MMBT adds (tanh GEMM epilogue, tanh backward, the generalized
text-embedding kernel, cross-entropy) against plain PyTorch fp32, and the registered `mmbt` model against the
fixture recorded from the real reference (tests/golden/mmbt_small64.npz) and the CPU oracle.
Tolerance: BASELINE.json north_star, 5e-2 for the bf16 path."""
import contextlib

import numpy as np
import pytest
import torch

from oracle import mmbt_oracle as O
from tests.golden_utils import load_mmbt_case
from tests.model_utils import build_mmbt, sample_to
from tests.test_kernels_gpu import close, nat, rnd, DEV
from mmf_amd.common.sample import SampleList

pytestmark = pytest.mark.gpu
TOL = 5e-2


def rel_err(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def test_gemm_tanh_epilogue_and_tanh_backward():
    M, N, K = 96, 768, 768
    A = rnd(M, K, scale=0.5); W = rnd(N, K, scale=0.05); bias = rnd(N, dtype=torch.float32)
    Y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    nat().gemm(A, W, Y, M, N, K, K, K, N, bias=bias, act=3)
    ref = torch.tanh(A.float() @ W.float().t() + bias)
    close(Y, ref, 1e-2, 1e-2, "tanh epilogue")
    dy = rnd(M, N); dx = torch.empty_like(dy)
    nat().tanh_bwd(dy, Y, dx)
    close(dx, dy.float() * (1 - Y.float() ** 2), 1e-2, 1e-3, "tanh bwd")
    # act = 4: GEMM output times (1 - aux^2)
    Z = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    nat().gemm(A, W, Z, M, N, K, K, K, N, act=4, aux=Y)
    close(Z, (A.float() @ W.float().t()) * (1 - Y.float() ** 2), 1e-2, 2e-2, "(1-aux^2) epilogue")


def test_embed_text_row_and_position_offsets():
    B, T, S, H, V = 3, 5, 16, 256, 500
    ids = torch.randint(0, V, (B, T), device=DEV); seg = torch.randint(0, 2, (B, T), device=DEV)
    word = rnd(V, H, dtype=torch.float32); pos = rnd(32, H, dtype=torch.float32); typ = rnd(2, H, dtype=torch.float32)
    y = torch.zeros(B * S, H, dtype=torch.bfloat16, device=DEV)
    nat().embed_text_fwd(ids, seg, word, pos, typ, y, B, T, S, H, 7, 3)
    ref = word[ids] + pos[torch.arange(T, device=DEV) + 3][None] + typ[seg]
    y3 = y.view(B, S, H)
    close(y3[:, 7:7 + T], ref, 1e-2, 1e-2, "embed rows 7.., positions 3..")
    assert float(y3[:, :7].abs().max()) == 0 and float(y3[:, 7 + T:].abs().max()) == 0
    # single-token form used for the modal start / end tokens
    tok = torch.randint(0, V, (B, 1), device=DEV); mt = torch.ones(B, 1, dtype=torch.int64, device=DEV)
    nat().embed_text_fwd(tok, mt, word, pos, typ, y, B, 1, S, H, 15, 15)
    close(y3[:, 15], word[tok[:, 0]] + pos[15] + typ[1], 1e-2, 1e-2, "end token row")


@pytest.mark.parametrize("B,C", [(16, 2), (64, 3129), (5, 7)])
def test_cross_entropy_forward_backward(B, C):
    import mmf_amd.functional as Fn
    s = rnd(B, C, dtype=torch.float32, scale=3.0).requires_grad_(True)
    t = torch.randint(0, C, (B,), device=DEV)
    t[B // 2] = -100
    loss = Fn.CrossEntropyFn.apply(s, t, -100)
    sr = s.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(sr, t, ignore_index=-100)
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    (loss * 1.7).backward(); (ref * 1.7).backward()
    close(s.grad, sr.grad, 1e-4, 1e-6, "dlogits")
    assert float(s.grad[B // 2].abs().max()) == 0


def test_dropout_fn_is_differentiable_with_the_same_mask():
    import mmf_amd.functional as Fn
    x = rnd(64, 768).requires_grad_(True)
    drop = nat().drop_cfg(0.1, 1234)
    y = Fn.DropoutFn.apply(x, drop)
    y.float().sum().backward()
    kept = y != 0
    assert 0.85 < kept.float().mean().item() < 0.95
    assert torch.equal(x.grad != 0, kept | ((x.detach() == 0) & (x.grad != 0)))
    close(x.grad[kept], torch.full_like(x.grad[kept], 1 / 0.9), 1e-2, 1e-3, "dropout grad scale")


# mmbt_decoder64: `is_decoder: true` (mmbt.py:244-272): the padding mask times a causal mask over modal + text positions — a materialised
# per-(query, key) mask for the attention kernels; the layers own the reference's unused crossattention blocks (no gradient)
@pytest.mark.parametrize("name", ["mmbt_small64", "mmbt_decoder64"])
def test_mmbt_golden_forward_loss_and_gradients(name):
    z, case, cfg, sd, sample = load_mmbt_case(name)
    model = build_mmbt(cfg, sd, O.SHARED)
    model.eval()
    seq = {}
    mm = model.model.bert.mmbt
    hook = mm.register_forward_hook(lambda m, i, o: seq.update(seq=o[0], pooled=o[1]))
    out = model(SampleList(sample_to(sample, "cuda")))
    hook.remove()
    np.testing.assert_allclose(out["scores"].detach().float().cpu().numpy(), z["scores"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(seq["seq"].detach().float().cpu().numpy(), z["sequence_output"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(seq["pooled"].detach().float().cpu().numpy(), z["pooled_output"], rtol=TOL, atol=TOL)
    (key, loss), = out["losses"].items()
    assert key == "train/hateful_memes/cross_entropy"
    assert abs(loss.item() - float(z["loss"])) <= TOL * abs(float(z["loss"]))
    loss.sum().backward()
    params = dict(model.named_parameters())
    worst = {}
    for gname, norm in zip(z["grad_names"], z["grad_norms"]):
        gname = str(gname)
        p = params[gname]
        if norm == 0.0 and "crossattention" in gname:      # built, never called (hf_layers.py:268-292)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, gname
            continue
        assert p.grad is not None, gname
        gn = float(p.grad.double().norm())
        if norm == 0.0:      # decoder mode: the classifier reads position 0, which sees only itself under the causal mask — a one-key softmax, so every
            assert gn <= 1e-6, (gname, gn)      # query / key gradient is exactly zero in the reference; here: differences of equal terms (~1e-10)
            continue
        if gname.endswith("self.key.bias"):   # identically zero in exact arithmetic: noise on both sides
            qn = float(params[gname.replace("key.bias", "query.bias")].grad.double().norm())
            assert gn <= TOL * qn + 1e-6, (gname, gn, qn)
            continue
        worst[gname] = abs(gn - norm) / norm
        full = "grad::" + gname
        if full in z.files:
            assert rel_err(p.grad, torch.from_numpy(z[full])) <= TOL, gname
    bad = {k: round(v, 4) for k, v in worst.items() if v > TOL}
    assert not bad, bad


def test_mmbt_all_gradients_match_oracle_without_modal_tokens():
    """Second layout (no start / end token, ragged masks, all-ones segments) against the pinned CPU oracle,
    every parameter's full gradient."""
    z, case, cfg, sd, sample = load_mmbt_case()
    cfg = dict(cfg, use_modal_start_token=False, use_modal_end_token=False)
    sample = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in sample.items()}
    sample["segment_ids"] = torch.ones_like(sample["segment_ids"])   # -> modal token type 0 (mmbt.py:398-404)
    model = build_mmbt(cfg, sd, O.SHARED)
    model.eval()
    out = model(SampleList(sample_to(sample, "cuda")))
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.mmbt_forward(sdr, cfg, {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in sample.items()})
    assert float((out["scores"].detach().float().cpu() - ref["scores"].detach()).abs().max()) <= TOL
    (key, loss), = out["losses"].items()
    ref_loss = O.cross_entropy(ref["scores"], sample["targets"])
    assert abs(loss.item() - ref_loss.item()) <= TOL * abs(ref_loss.item())
    loss.sum().backward(); ref_loss.backward()
    params = dict(model.named_parameters())
    errs = {}
    for k, v in sdr.items():
        p = params["model." + k]
        assert p.grad is not None and v.grad is not None, k
        if k.endswith("self.key.bias"):
            continue
        errs[k] = rel_err(p.grad, v.grad)
    bad = {k: round(e, 4) for k, e in errs.items() if e > TOL}
    assert not bad, bad


@pytest.mark.parametrize("fp32", [False, True])
def test_mmbt_per_position_modal_token_type_ids_match_oracle(fp32):
    """A caller's own `modal_token_type_ids` [B, L] (ModalEmbeddings.forward looks every position up, mmbt.py:117-127; MMBTBase only ever builds a
    constant block): start token, every projected feature and the end token carry their own type row, forward and backward (the token-type table
    collects each row's gradient) — against the pinned oracle, on the bf16 kernels and under mmf_amd.fp32_training()."""
    import mmf_amd
    z, case, cfg, sd, sample = load_mmbt_case()
    sample = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in sample.items()}
    B = sample["input_ids"].shape[0]
    L = sample["image_feature_0"].shape[1] + 2
    g = torch.Generator().manual_seed(5)
    sample["modal_token_type_ids"] = torch.randint(0, 2, (B, L), generator=g)
    model = build_mmbt(cfg, sd, O.SHARED)
    model.eval()
    tol = 1e-3 if fp32 else TOL
    with (mmf_amd.fp32_training() if fp32 else contextlib.nullcontext()):
        out = model(SampleList(sample_to(sample, "cuda")))
        (key, loss), = out["losses"].items()
        loss.sum().backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.mmbt_forward(sdr, cfg, {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in sample.items()})
    assert float((out["scores"].detach().float().cpu() - ref["scores"].detach()).abs().max()) <= tol
    ref_loss = O.cross_entropy(ref["scores"], sample["targets"])
    assert abs(loss.item() - ref_loss.item()) <= tol * abs(ref_loss.item())
    ref_loss.backward()
    # a constant block gives another answer: the per-position ids are really used
    const = O.mmbt_forward(sd, cfg, {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in sample.items() if k != "modal_token_type_ids"})
    assert float((const["scores"] - ref["scores"].detach()).abs().max()) > 10 * tol * 0 + 1e-4
    params = dict(model.named_parameters())
    bad = {}
    for k, v in sdr.items():
        p = params["model." + k]
        assert p.grad is not None and v.grad is not None, k
        if k.endswith("self.key.bias"):
            continue
        e = rel_err(p.grad, v.grad)
        if e > tol:
            bad[k] = round(e, 5)
    assert not bad, bad
    with pytest.raises(ValueError):
        sample["modal_token_type_ids"] = torch.zeros(B, L + 1, dtype=torch.long)
        model(SampleList(sample_to(sample, "cuda")))


def test_mmbt_training_step_is_seed_reproducible_and_updates():
    import mmf_amd
    from mmf_amd.modules.optimizers import AdamW
    z, case, cfg, sd, sample = load_mmbt_case()
    model = build_mmbt(cfg, sd, O.SHARED)
    model.train()
    batch = sample_to(sample, "cuda")
    torch.manual_seed(3)
    a = model(SampleList(dict(batch)))["scores"].float().clone()
    torch.manual_seed(3)
    b = model(SampleList(dict(batch)))["scores"].float().clone()
    assert torch.equal(a, b)
    from mmf_amd.utils.configuration import Config
    full = Config(model="mmbt", optimizer=dict(params=dict(lr=1e-3)), model_config=dict(mmbt=model.config))
    opt = AdamW(model.get_optimizer_parameters(full), lr=1e-3)
    losses = []
    for _ in range(8):
        out = model(SampleList(dict(batch)))
        loss = sum(v.sum() for v in out["losses"].values())
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0], losses


def test_mmbt_with_trainable_fc7_modal_encoder_matches_oracle():
    """projects/hateful_memes/configs/mmbt/with_features.yaml: `modal_encoder: finetune_faster_rcnn_fpn_fc7` (mmf/modules/encoders.py:
    116-180, relu(lc(x)) on the pre-extracted features, trainable) ahead of the modal projection.  The oracle takes the encoded
    features as a non-leaf tensor, so autograd carries its loss back into `lc`: scores, loss and the encoder's gradients must agree."""
    z, case, cfg, sd, sample = load_mmbt_case()
    D = cfg["modal_hidden_size"]
    from tests.model_utils import mmbt_model_config
    from mmf_amd.utils.build import build_model
    import warnings
    mc = mmbt_model_config(cfg, modal_encoder=dict(type="finetune_faster_rcnn_fpn_fc7", params=dict(
        in_dim=D, out_dim=D, weights_file="absent_w.pkl", bias_file="absent_b.pkl")))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = build_model(mc)
    g = torch.Generator().manual_seed(5)
    lc_w = torch.randn(D, D, generator=g) * 0.1
    lc_b = torch.randn(D, generator=g) * 0.1
    full = {"model." + k: v for k, v in sd.items()}
    for alias, src in O.SHARED.items():
        full["model." + alias] = sd[src]
    full["model.bert.mmbt.modal_encoder.encoder.lc.weight"] = lc_w
    full["model.bert.mmbt.modal_encoder.encoder.lc.bias"] = lc_b
    model.load_state_dict(full, strict=True)
    model = model.to("cuda")
    model.eval()
    out = model(SampleList(sample_to(sample, "cuda")))
    (key, loss), = out["losses"].items()
    loss.backward()
    wr, br = lc_w.clone().requires_grad_(True), lc_b.clone().requires_grad_(True)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    s2 = dict(sample)
    s2["image_feature_0"] = torch.relu(torch.nn.functional.linear(sample["image_feature_0"], wr, br))
    ref = O.mmbt_forward(sdr, cfg, s2, train=False)
    ref_loss = O.cross_entropy(ref["scores"], sample["targets"])
    ref_loss.backward()
    assert (out["scores"].float().cpu() - ref["scores"]).abs().max().item() <= TOL
    assert abs(loss.item() - ref_loss.item()) <= TOL * abs(ref_loss.item())
    lc = model.model.bert.mmbt.modal_encoder.encoder.lc
    assert lc.weight.grad is not None and lc.bias.grad is not None
    assert rel_err(lc.weight.grad, wr.grad) <= 6e-2, rel_err(lc.weight.grad, wr.grad)
    assert rel_err(lc.bias.grad, br.grad) <= 6e-2, rel_err(lc.bias.grad, br.grad)
    pw = model.model.bert.mmbt.modal_encoder.proj_embeddings.weight
    assert rel_err(pw.grad, sdr["bert.mmbt.modal_encoder.proj_embeddings.weight"].grad) <= TOL
