"""Host logic of the fp32-accurate forward path (mmf_amd/fp32_path.py) on CPU: with the kernel wrappers replaced by extent /
dtype checkers (tests/native_stub.py), a VisualBERT forward inside `mmf_amd.fp32_inference()` must route EVERY operator to the
fp32 entry points (no bf16 kernel, no shadow cast), keep activations fp32, refuse training-mode dropout, and leave the regular
bf16 routing untouched outside the block.  Numbers are the `-m gpu` tests' job (tests/test_fp32_path_gpu.py)."""
import pytest
import torch

import mmf_amd
from mmf_amd import fp32_path
from mmf_amd.common.sample import SampleList
from tests import golden_utils as G
from tests import model_utils as MU
from tests import native_stub

FP32_CALLS = {"gemm_f32", "attention_f32_fwd", "layernorm_f32_fwd", "embed_text_f32_fwd", "gather_rows_f32", "make_additive_mask", "visual_masks",
              "bce_logits_fwd", "cross_entropy_fwd", "copy_rows"}


def _names(calls):
    return {c[0] for c in calls}


def test_visual_bert_forward_routes_to_the_fp32_kernels_only():
    z, case, cfg, sd, sample = G.load_case("small64")
    model = MU.build_visual_bert(cfg, sd, device="cpu", output_hidden_states=True)
    model.eval()
    with native_stub.installed() as calls:
        with mmf_amd.fp32_inference():
            assert fp32_path.active() and not torch.is_grad_enabled()
            out = model(SampleList(sample))
        assert not fp32_path.active() and torch.is_grad_enabled()
        names = _names(calls)
        assert names <= FP32_CALLS, names - FP32_CALLS
        L = cfg["num_hidden_layers"]
        # per layer: packed Q|K|V, out-proj, FFN-up, FFN-down; plus the visual projection, the head transform and the classifier
        assert sum(c[0] == "gemm_f32" for c in calls) == 4 * L + 3
        assert sum(c[0] == "attention_f32_fwd" for c in calls) == L
        assert sum(c[0] == "layernorm_f32_fwd" for c in calls) == 2 * L + 2
    assert out["scores"].dtype == torch.float32 and out["sequence_output"].dtype == torch.float32
    assert out["scores"].shape == tuple(z["scores"].shape)
    assert len(out["losses"]) == 1


def test_nlvr2_head_and_pooler_route_to_the_fp32_kernels():
    z, case, cfg, sd, sample = G.load_nlvr2_case()
    model = MU.build_visual_bert(cfg, sd, device="cpu", training_head_type="nlvr2", pooler_strategy="default",
                                 losses=[dict(type="cross_entropy")])
    model.eval()
    with native_stub.installed() as calls, mmf_amd.fp32_inference():
        out = model(SampleList(sample))
        assert _names(calls) <= FP32_CALLS, _names(calls) - FP32_CALLS
    assert out["scores"].dtype == torch.float32 and out["scores"].shape == tuple(z["scores"].shape)


def test_mmbt_and_mmft_route_to_the_fp32_kernels():
    from oracle.mmbt_oracle import SHARED
    from oracle.mmft_oracle import shared
    z, case, cfg, sd, sample = G.load_mmbt_case()
    model = MU.build_mmbt(cfg, sd, SHARED, device="cpu")
    model.eval()
    with native_stub.installed() as calls, mmf_amd.fp32_inference():
        out = model(SampleList(sample))
        assert _names(calls) <= FP32_CALLS, _names(calls) - FP32_CALLS
    assert out["scores"].dtype == torch.float32 and out["scores"].shape == tuple(z["scores"].shape)
    z, case, cfg, sd, sample = G.load_mmft_case()
    model = MU.build_mmft(cfg, sd, shared(cfg), device="cpu")
    model.eval()
    with native_stub.installed() as calls, mmf_amd.fp32_inference():
        out = model(SampleList(sample))
        assert _names(calls) <= FP32_CALLS | {"rows_add_embed_f32"}, _names(calls) - FP32_CALLS
        assert any(c[0] == "rows_add_embed_f32" for c in calls)
    assert out["scores"].dtype == torch.float32 and out["scores"].shape == tuple(z["scores"].shape)


def test_training_mode_dropout_is_refused_and_bf16_routing_is_untouched_outside():
    z, case, cfg, sd, sample = G.load_case("small64")
    model = MU.build_visual_bert(cfg, sd, device="cpu")
    model.train()
    with native_stub.installed():
        with mmf_amd.fp32_inference():
            with pytest.raises(RuntimeError, match="forward-only"):
                model(SampleList(sample))
    assert not fp32_path.active()
    model.eval()
    with native_stub.installed() as calls:
        model(SampleList(sample))
        names = _names(calls)
        assert "gemm" in names and "attention_fwd" in names and not (names & {"gemm_f32", "attention_f32_fwd", "layernorm_f32_fwd", "embed_text_f32_fwd", "gather_rows_f32"}), names


def test_unsupported_shapes_fail_loudly():
    x = torch.zeros(2, 8, 96)
    w = torch.zeros(96, 96); b = torch.zeros(96); g = torch.ones(96)
    with native_stub.installed(), mmf_amd.fp32_inference():
        with pytest.raises(NotImplementedError, match="head_dim 64 and 128"):
            fp32_path.transformer_layer(x, w, b, w, b, w, b, w, b, g, b, w, b, w, b, g, b, None, 3, 1e-12, 1e-12)
        with pytest.raises(TypeError, match="float32"):
            fp32_path.layer_norm(x.bfloat16(), g, b, 1e-12)


@pytest.mark.parametrize("name", ["vilbert_small", "vilbert_dyn", "vilbert_nlvr2"])
def test_vilbert_routes_to_the_fp32_kernels_only(name):
    """Every kernel a ViLBERT forward launches inside fp32_inference() is an fp32 one (no bf16 GEMM / attention / LayerNorm)."""
    z, case, cfg, sd, sample = G.load_vilbert_case(name)
    over = dict(training_head_type="nlvr2", losses=[dict(type="cross_entropy")]) if name == "vilbert_nlvr2" else {}
    model = MU.build_vilbert(cfg, sd, device="cpu", **over)
    model.eval()
    extra = {"pad_rows_f32", "eltwise_f32", "masked_mean_f32", "rowgroup_scale_f32", "gate_sigmoid_fwd"}
    with native_stub.installed() as calls, mmf_amd.fp32_inference():
        out = model(SampleList(sample))
        assert _names(calls) <= FP32_CALLS | extra, _names(calls) - FP32_CALLS - extra
        assert any(c[0] == "attention_f32_fwd" and c[3] != c[4] for c in calls)          # the co-attention: Sq != Sk
        assert ("rowgroup_scale_f32" in _names(calls)) == (name == "vilbert_dyn")
    assert out["scores"].dtype == torch.float32 and out["scores"].shape == tuple(z["scores"].shape)


def test_uniter_routes_to_the_fp32_kernels_only():
    z, case, cfg, sd, sample = G.load_uniter_case()
    model = MU.build_uniter(cfg, sd, device="cpu")
    model.eval()
    extra = {"pad_rows_f32", "eltwise_f32", "rows_add_embed_f32"}
    with native_stub.installed() as calls, mmf_amd.fp32_inference():
        out = model(SampleList(sample))
        assert _names(calls) <= FP32_CALLS | extra, _names(calls) - FP32_CALLS - extra
    assert out["scores"].dtype == torch.float32 and out["scores"].shape == tuple(z["scores"].shape)


def test_m4c_routes_to_the_fp32_kernels_only():
    """M4C inside fp32_inference(): teacher-forcing pass and the greedy decoding loop (the reference's re-encoding loop: the K|V-cached decoder
    is a bf16-path optimisation) launch fp32 kernels only; the prefix-LM tail reaches the fp32 attention."""
    z, case, cfg, sd, sample = G.load_m4c_case()
    model = MU.build_m4c(cfg, sd, device="cpu")
    model.eval()
    extra = {"pad_rows_f32", "eltwise_f32", "rows_add_embed_f32", "l2norm_rows_f32", "gather_rows2_f32", "ptr_scores_f32", "bce_rowmask_fwd"}
    with native_stub.installed() as calls, mmf_amd.fp32_inference():
        out = model(SampleList(sample))
        assert _names(calls) <= FP32_CALLS | extra, _names(calls) - FP32_CALLS - extra
        assert {"l2norm_rows_f32", "gather_rows2_f32", "ptr_scores_f32"} <= _names(calls)
    assert out["scores"].dtype == torch.float32 and out["scores"].shape == tuple(z["decode_scores"].shape)
