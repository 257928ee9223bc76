"""Eager == TorchScript on the HIP path, the reference's own check (tests/models/test_visual_bert.py:43-49 ->
tests/test_utils.py:270-285: random ids [1, 128], features [1, 100, 2048], `torch.allclose` on the scores), plus what the
custom-op boundary promises: gradients flow through the scripted module and match the eager ones, the scripted module survives
`torch.jit.save` / `load`, and the ops refuse host tensors."""
import io

import pytest
import torch

from oracle import visual_bert_oracle as O
from tests.golden_utils import load_case
from tests.model_utils import build_visual_bert, sample_to
from mmf_amd.common.sample import SampleList

pytestmark = pytest.mark.gpu


def _tensors_only(sample):
    return {k: v for k, v in sample.items() if isinstance(v, torch.Tensor)}


def test_scripted_visual_bert_equals_eager_like_the_reference_test():
    cfg = dict(O.DEFAULT_CONFIG)
    cfg["num_hidden_layers"] = 2
    cfg["num_labels"] = 2
    model = build_visual_bert(cfg, O.init_state_dict(cfg, seed=3)).eval()
    g = torch.Generator().manual_seed(0)
    sample = {"input_ids": torch.randint(0, 30255, (1, 128), generator=g), "input_mask": torch.ones(1, 128, dtype=torch.long),
              "segment_ids": torch.zeros(1, 128, dtype=torch.long), "image_feature_0": torch.rand(1, 100, 2048, generator=g)}
    batch = sample_to(sample, "cuda")
    with torch.no_grad():
        eager = model(SampleList(dict(batch)))["scores"]
    scripted = torch.jit.script(model)
    with torch.no_grad():
        out = scripted(dict(batch))["scores"]
    assert torch.allclose(eager, out)
    assert torch.equal(eager, out)                     # same kernels, same launches: bit-identical, not just close
    buf = io.BytesIO()
    torch.jit.save(scripted, buf)
    buf.seek(0)
    loaded = torch.jit.load(buf, map_location="cuda")
    with torch.no_grad():
        again = loaded(dict(batch))["scores"]
    assert torch.equal(again, eager)


def test_gradients_flow_through_the_scripted_model_and_match_eager():
    z, case, cfg, sd, sample = load_case("small64")
    model = build_visual_bert(cfg, sd).eval()
    batch = sample_to(_tensors_only(sample), "cuda")
    batch["image_feature_0"] = batch["image_feature_0"].contiguous()
    out = model(SampleList(dict(batch, targets=sample["targets"].cuda(), dataset_name="vqa2", dataset_type="train")))
    out["scores"].float().square().sum().backward()
    ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    scripted = torch.jit.script(model)
    s = scripted(dict(batch))["scores"]
    s.float().square().sum().backward()
    got = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(got) == set(ref) and len(ref) > 30
    for n in ref:
        assert torch.equal(got[n], ref[n]), n


def test_ops_refuse_host_tensors_and_wrong_dtypes():
    from mmf_amd._native import NativeLibraryError
    x = torch.randn(4, 256)
    with pytest.raises((NativeLibraryError, RuntimeError)):
        torch.ops.mmf_amd.layer_norm(x, torch.ones(256), torch.zeros(256), 1e-12)
    xc = torch.randn(4, 256, device="cuda")
    with pytest.raises((NativeLibraryError, RuntimeError)):      # gamma must be fp32 in HBM
        torch.ops.mmf_amd.layer_norm(xc, torch.ones(256, device="cuda", dtype=torch.float16), torch.zeros(256, device="cuda"), 1e-12)
    y = torch.ops.mmf_amd.layer_norm(xc, torch.ones(256, device="cuda"), torch.zeros(256, device="cuda"), 1e-12)
    ref = torch.nn.functional.layer_norm(xc.bfloat16().float(), (256,))
    assert float((y.float() - ref).abs().max()) < 3e-2


_STANDALONE = r"""
import sys, torch
torch.ops.load_library(sys.argv[1])            # libmmf_amd_ops.so (finds libmmf_amd.so beside it); the Python package is NOT imported
assert "mmf_amd" not in sys.modules
m = torch.jit.load(sys.argv[2], map_location="cuda")
blob = torch.load(sys.argv[3])
batch = {k: v.cuda() for k, v in blob["batch"].items()}
with torch.no_grad():
    out = m(batch)["scores"]
assert torch.equal(out.cpu(), blob["scores"]), float((out.cpu().float() - blob["scores"].float()).abs().max())
# training through the loaded module: C++ autograd nodes, gradients for the parameters the scripted module carries
m.train()
torch.manual_seed(5)
s = m(batch)["scores"]
s.float().square().sum().backward()
n = sum(1 for p in m.parameters() if p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0)
assert n > 30, n
try:
    torch.ops.mmf_amd.layer_norm(torch.randn(4, 256), torch.ones(256), torch.zeros(256), 1e-12)
    raise SystemExit("a host tensor was accepted")
except RuntimeError as e:
    assert "HBM" in str(e) or "CPU" in str(e), e
print("standalone ok", n)
"""


def test_saved_scripted_model_runs_after_load_library_alone(tmp_path):
    """SURVEY section 8(b), last row: the operator ABI is a shared library loaded with torch.ops.load_library.  A scripted VisualBERT saved here
    runs in a FRESH interpreter that never imports the Python package — forward bit-identical, backward through the C++ autograd nodes."""
    import subprocess
    import sys

    from mmf_amd import _ops_native
    if not _ops_native.NATIVE:
        pytest.skip("MMF_AMD_PY_OPS=1: the operators are declared from Python in this process")
    cfg = dict(O.DEFAULT_CONFIG)
    cfg["num_hidden_layers"] = 2
    cfg["num_labels"] = 16
    model = build_visual_bert(cfg, O.init_state_dict(cfg, seed=4)).eval()
    g = torch.Generator().manual_seed(1)
    sample = {"input_ids": torch.randint(0, 30255, (2, 128), generator=g), "input_mask": torch.ones(2, 128, dtype=torch.long),
              "segment_ids": torch.zeros(2, 128, dtype=torch.long), "image_feature_0": torch.rand(2, 100, 2048, generator=g)}
    batch = sample_to(sample, "cuda")
    with torch.no_grad():
        eager = model(SampleList(dict(batch)))["scores"]
    scripted = torch.jit.script(model)
    mp, bp, sp = str(tmp_path / "vb.pt"), str(tmp_path / "io.pt"), str(tmp_path / "standalone.py")
    torch.jit.save(scripted, mp)
    torch.save({"batch": {k: v.cpu() for k, v in batch.items()}, "scores": eager.cpu()}, bp)
    open(sp, "w").write(_STANDALONE)
    r = subprocess.run([sys.executable, sp, _ops_native.OPS_LIB_PATH, mp, bp], capture_output=True, text=True, timeout=280, cwd=str(tmp_path))
    assert r.returncode == 0 and "standalone ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_native_library_argument_errors_are_torch_checks():
    from mmf_amd import _ops_native
    if not _ops_native.NATIVE:
        pytest.skip("MMF_AMD_PY_OPS=1")
    x = torch.randn(2, 8, 768, device="cuda")
    w = torch.randn(768, 768, device="cuda")
    with pytest.raises(RuntimeError, match="must be float|Float"):          # fp16 bias
        torch.ops.mmf_amd.linear(x, w, torch.zeros(768, device="cuda", dtype=torch.float16), False)
    with pytest.raises(RuntimeError, match="weight must be"):
        torch.ops.mmf_amd.linear(x, torch.randn(768, 512, device="cuda"), None, False)
    with pytest.raises(RuntimeError, match="B, S, H"):
        torch.ops.mmf_amd.gather_rows(torch.randn(4, 768, device="cuda"), torch.zeros(4, dtype=torch.long, device="cuda"), 0.0, False)


def test_visual_masks_op_matches_the_unfused_massaging():
    """torch.ops.mmf_amd.visual_masks = the arange / compare / zeros_like / cat / additive-mask / sum - 2 sequence of VisualBERT.forward's input
    massaging (visual_bert.py:444-467, 525-556, 389-392) in one launch."""
    g = torch.Generator().manual_seed(0)
    B, T, R = 5, 128, 100
    input_mask = (torch.rand(B, T, generator=g) < 0.8).long()
    input_mask[:, 0] = 1
    dims = torch.tensor([100, 73, 1, 0, 55])
    for image_dim in (dims, dims.view(B, 1), None):
        out = torch.ops.mmf_amd.visual_masks(input_mask.cuda(), None if image_dim is None else image_dim.cuda(), R)
        image_mask, attention_mask, vtype, mask_add, pool = (t.cpu() for t in out)
        d = torch.full((B, 1), R) if image_dim is None else image_dim.view(B, 1)
        want_im = (torch.arange(R).expand(B, R) < d).long()
        want_am = torch.cat((input_mask, want_im), dim=-1)
        assert torch.equal(image_mask, want_im) and torch.equal(attention_mask, want_am) and attention_mask.dtype == torch.int64
        assert torch.equal(vtype, torch.zeros_like(want_im))
        assert torch.equal(mask_add, (1.0 - want_am.float()) * -10000.0) and mask_add.dtype == torch.float32
        assert torch.equal(pool, input_mask.sum(1) - 2)
    with pytest.raises(RuntimeError, match="HBM"):
        torch.ops.mmf_amd.visual_masks(input_mask, None, R)
