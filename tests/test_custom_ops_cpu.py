"""The PyTorch custom-op surface (mmf_amd/ops.py): operators registered with `torch.library` under
`torch.ops.mmf_amd`, and the registered VisualBERT compiling under `torch.jit.script` into a graph that calls them — the host-side
half of the reference's scriptability tests (tests/models/test_visual_bert.py:40-49); the numerical half runs on the GPU
(tests/test_custom_ops_gpu.py)."""
import io

import pytest
import torch

from tests.golden_utils import load_case
from tests.model_utils import build_visual_bert


def test_operators_are_registered_with_schemas():
    from mmf_amd import ops
    sch = ops.schemas()
    for name in ("transformer_layer", "visio_linguistic_embeddings", "additive_mask", "gather_rows", "dense_gelu", "layer_norm", "linear",
                 "linear_tanh", "dropout", "pair_halves"):
        assert name in sch
        op = getattr(torch.ops.mmf_amd, name)
        parsed = op.default._schema
        assert parsed.name == "mmf_amd::" + name
        assert str(parsed.returns[0].type) == "Tensor"
    s = torch.ops.mmf_amd.transformer_layer.default._schema
    names = [a.name for a in s.arguments]
    assert names[:3] == ["x", "wq", "bq"] and "mask_add" in names and names[-1] == "causal_tail"
    assert str(next(a for a in s.arguments if a.name == "mask_add").type) == "Optional[Tensor]"


def test_there_is_no_cpu_path_behind_the_ops():
    from mmf_amd._native import NativeLibraryError
    x = torch.randn(4, 8)
    with pytest.raises((NativeLibraryError, RuntimeError)):
        torch.ops.mmf_amd.layer_norm(x, torch.ones(8), torch.zeros(8), 1e-12)


def test_visual_bert_scripts_into_a_graph_of_mmf_amd_ops():
    z, case, cfg, sd, sample = load_case("small64")
    model = build_visual_bert(cfg, sd, device="cpu").eval()
    scripted = torch.jit.script(model)
    g = str(scripted.inlined_graph)
    for op in ("mmf_amd::transformer_layer", "mmf_amd::visio_linguistic_embeddings", "mmf_amd::additive_mask", "mmf_amd::gather_rows",
               "mmf_amd::dense_gelu", "mmf_amd::layer_norm", "mmf_amd::linear"):
        assert op in g, op
    assert g.count("mmf_amd::transformer_layer") == cfg["num_hidden_layers"]
    # the parameter tree survives scripting and a save / load round trip under the reference's names
    buf = io.BytesIO()
    torch.jit.save(scripted, buf)
    buf.seek(0)
    loaded = torch.jit.load(buf)
    keys = set(model.state_dict().keys())
    assert set(scripted.state_dict().keys()) == keys == set(loaded.state_dict().keys())
    assert "model.bert.encoder.layer.1.attention.self.query.weight" in keys
    for k, v in model.state_dict().items():
        assert torch.equal(loaded.state_dict()[k], v)
