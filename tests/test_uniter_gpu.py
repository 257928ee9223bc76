"""UNITER on the HIP path (GPU): the kernels / autograd pieces it adds against plain PyTorch fp32, LayerNorm at the head widths
UNITER uses (1536) and beyond (2048), and the registered `uniter` model against the fixture recorded from the real reference
(tests/golden/uniter_small64.npz) and the pinned CPU oracle.  Tolerance 5e-2 (bf16 path)."""
import numpy as np
import pytest
import torch

from oracle import uniter_oracle as O
from oracle.visual_bert_oracle import logit_bce
from tests.golden_utils import load_uniter_case
from tests.model_utils import build_uniter, sample_to
from tests.test_kernels_gpu import close, nat, rnd, DEV
from mmf_amd.common.sample import SampleList

pytestmark = pytest.mark.gpu
TOL = 5e-2


def rel_err(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def test_feature_table_add_small_k_linear_and_add():
    import mmf_amd.functional as Fn
    B, R, D, H = 3, 7, 72, 128
    feats = torch.randn(B, R, D, device=DEV)
    mask = (torch.rand(B, R, device=DEV) > 0.3).long()
    table = torch.randn(2, D, device=DEV).requires_grad_(True)
    with torch.no_grad():
        table[0].zero_()
    y = Fn.FeatureTableAddFn.apply(feats, mask, table, 0)
    close(y, feats + table.detach()[mask], 1e-2, 1e-2, "features + mask embedding")
    g = rnd(B, R, D, seed=3)
    y.backward(g)
    assert float(table.grad[0].abs().max()) == 0.0
    close(table.grad[1], (g.float() * mask[..., None]).sum((0, 1)), 1e-4, 1e-3, "mask embedding gradient")
    # 7-wide linear
    x = torch.rand(B, R, 7, device=DEV) * 300
    w = (torch.randn(H, 7, device=DEV) * 0.01).requires_grad_(True); b = torch.randn(H, device=DEV).requires_grad_(True)
    out = Fn.SmallKLinearFn.apply(x, w, b)
    ref = x.bfloat16().float() @ w.detach().bfloat16().float().t() + b.detach()
    close(out, ref, 1e-2, 2e-2, "7-wide linear")
    go = rnd(B, R, H, seed=4)
    out.backward(go)
    close(w.grad, go.float().reshape(-1, H).t() @ x.bfloat16().float().reshape(-1, 7), 1e-2, 1e-1, "7-wide wgrad")
    close(b.grad, go.float().sum((0, 1)), 1e-3, 1e-2, "7-wide bias grad")
    a1 = rnd(B, R, H, seed=5).requires_grad_(True); a2 = rnd(B, R, H, seed=6).requires_grad_(True)
    s = Fn.AddFn.apply(a1, a2)
    close(s, a1.detach().float() + a2.detach().float(), 1e-2, 1e-2, "add")
    s.backward(go)
    assert torch.equal(a1.grad, go) and torch.equal(a2.grad, go)


@pytest.mark.parametrize("rows,H", [(32, 1536), (9, 2048), (100, 1280)])
def test_layernorm_wide_rows_forward_backward(rows, H):
    import mmf_amd.functional as Fn
    x = rnd(rows, H).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(H, device=DEV)).requires_grad_(True); beta = (0.1 * torch.randn(H, device=DEV)).requires_grad_(True)
    y = Fn.LayerNormFn.apply(x, gamma, beta, 1e-6)
    xr = x.detach().float().requires_grad_(True); gr = gamma.detach().clone().requires_grad_(True); br = beta.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (H,), gr, br, 1e-6)
    close(y, ref, 1e-2, 2e-2, "wide LN forward")
    g = rnd(rows, H, seed=9)
    y.backward(g); ref.backward(g.float())
    assert rel_err(x.grad, xr.grad) <= 1e-2 and rel_err(gamma.grad, gr.grad) <= 1e-2 and rel_err(beta.grad, br.grad) <= 1e-2


def _grad_check(model, out, ref, ref_loss, sdr, prefix=""):
    assert rel_err(out["scores"], ref["scores"]) <= TOL
    (key, loss), = out["losses"].items()
    assert key == "train/vqa2/logit_bce"
    assert abs(loss.sum().item() - ref_loss.item()) <= TOL * abs(ref_loss.item())
    loss.sum().backward(); ref_loss.backward()
    params = dict(model.named_parameters())
    errs = {}
    for k, v in sdr.items():
        p = params[k]
        if v.grad is None or float(v.grad.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        if k.endswith(".key.bias"):
            continue
        errs[k] = rel_err(p.grad, v.grad)
    return {k: round(e, 4) for k, e in errs.items() if e > TOL}


def test_uniter_golden_forward_loss_and_gradients():
    z, case, cfg, sd, sample = load_uniter_case()
    model = build_uniter(cfg, sd)
    model.eval()
    got = {}
    hook = model.uniter.uniter.register_forward_hook(lambda m, i, o: got.update(seq=o[0]))
    out = model(SampleList(sample_to(sample, "cuda")))
    hook.remove()
    np.testing.assert_allclose(out["scores"].detach().float().cpu().numpy(), z["scores"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(got["seq"].detach().float().cpu().numpy(), z["sequence_output"], rtol=TOL, atol=TOL)
    (key, loss), = out["losses"].items()
    assert key == str(z["loss_key"]) and abs(loss.sum().item() - float(z["loss"])) <= TOL * abs(float(z["loss"]))
    loss.sum().backward()
    params = dict(model.named_parameters())
    worst = {}
    for gname, norm in zip(z["grad_names"], z["grad_norms"]):
        gname = str(gname)
        p = params[gname]
        if norm == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, gname
            continue
        assert p.grad is not None, gname
        if gname.endswith(".key.bias"):
            continue
        worst[gname] = abs(float(p.grad.double().norm()) - norm) / norm
        full = "grad::" + gname
        if full in z.files:
            assert rel_err(p.grad, torch.from_numpy(z[full])) <= TOL, gname
    bad = {k: round(v, 4) for k, v in worst.items() if v > TOL}
    assert not bad, bad
    mg = params["uniter.uniter.img_embeddings.mask_embedding.weight"].grad
    assert float(mg[0].abs().max()) == 0.0 and float(mg[1].abs().max()) > 0.0


def test_uniter_default_head_width_and_normalised_boxes_match_oracle():
    """Head hidden width 1536 (the reference's default `mlp` head for UNITER: LayerNorm over 1536 columns), boxes given as
    fractions (the reference then divides by the image size, uniter.py:697-711), every parameter's full gradient."""
    z, case, cfg, sd, sample = load_uniter_case()
    cfg = dict(cfg, head_hidden_size=1536, num_labels=29)
    g = torch.Generator().manual_seed(9)
    sd = dict(sd)
    for k, shp in O.parameter_shapes(cfg).items():
        if k not in sd or tuple(sd[k].shape) != tuple(shp):
            sd[k] = (1.0 + 0.05 * torch.randn(shp, generator=g)) if k.endswith("LayerNorm.weight") else 0.03 * torch.randn(shp, generator=g)
    B, T, R = 2, 16, 12
    ids = torch.randint(1, cfg["vocab_size"], (B, T), generator=g)
    mask = torch.ones(B, T, dtype=torch.long); mask[1, 9:] = 0; ids[mask == 0] = 0
    xy = torch.rand(B, R, 4, generator=g) * 0.45
    bbox = torch.stack([xy[..., 0], xy[..., 1], xy[..., 0] + xy[..., 2] + 0.05, xy[..., 1] + xy[..., 3] + 0.05], dim=-1)
    targets = torch.zeros(B, cfg["num_labels"]); targets[0, 3] = 1.0; targets[1, 11] = 0.6
    sample = {"input_ids": ids, "input_mask": mask, "segment_ids": torch.zeros(B, T, dtype=torch.long),
              "image_feature_0": torch.randn(B, R, cfg["img_dim"], generator=g),
              "image_info_0": {"max_features": torch.tensor([12, 8]), "bbox": bbox, "image_width": torch.tensor([640, 500]),
                               "image_height": torch.tensor([480, 375])},
              "targets": targets, "dataset_name": "vqa2", "dataset_type": "train"}
    model = build_uniter(cfg, sd)
    model.eval()
    out = model(SampleList(sample_to(sample, "cuda")))
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.uniter_forward(sdr, cfg, dict(sample))
    bad = _grad_check(model, out, ref, logit_bce(ref["scores"], targets), sdr)
    assert not bad, bad
