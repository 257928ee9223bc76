"""Host logic of `mmf_amd.fp32_training()` without a GPU: the kernel wrappers are replaced by recorders (tests/native_stub.py), the
VisualBERT step runs forward and backward through the fp32 autograd nodes, and what is checked is the plumbing — every launch is an
fp32 kernel, every parameter receives an fp32 gradient of its own shape, operators without a backward refuse."""
import pytest
import torch

import mmf_amd
from mmf_amd.common.sample import SampleList
from tests import golden_utils as G
from tests import model_utils as MU
from tests import native_stub

BF16_KERNELS = {"gemm", "gemm_grouped", "attention_fwd", "attention_bwd", "layernorm_fwd", "layernorm_bwd", "embed_text_fwd", "dropout",
                "gather_rows", "bce_logits_bwd", "cast_f32_to_bf16"}


@pytest.mark.parametrize("train", [False, True])
def test_visual_bert_step_builds_an_fp32_graph_only(train):
    z, case, cfg, sd, sample = G.load_case("small64")
    model = MU.build_visual_bert(cfg, sd, device="cpu")
    model.train(train)
    with native_stub.installed() as calls:
        with mmf_amd.fp32_training():
            out = model(SampleList(sample))
        (key, loss), = out["losses"].items()
        assert out["scores"].dtype == torch.float32 and loss.requires_grad
        loss.backward()
        names = {c[0] for c in calls}
    assert not (names & BF16_KERNELS), names & BF16_KERNELS
    assert {"gemm_f32", "attention_f32_fwd", "attention_f32_bwd", "layernorm_f32_fwd_stats", "layernorm_f32_bwd", "colsum_f32",
            "scatter_add_rows_f32", "bce_logits_f32_bwd"} <= names, names
    assert ("dropout_f32" in names) == train
    for k, p in model.named_parameters():
        if "pooler" in k:            # computed and discarded under pooler_strategy == "vqa": no gradient, as in the reference
            assert p.grad is None, k
            continue
        assert p.grad is not None and p.grad.dtype == torch.float32 and p.grad.shape == p.shape, k


def test_nlvr2_step_builds_an_fp32_graph():
    z, case, cfg, sd, sample = G.load_nlvr2_case()
    model = MU.build_visual_bert(cfg, sd, device="cpu", training_head_type="nlvr2", pooler_strategy="default", losses=[dict(type="cross_entropy")])
    model.eval()
    with native_stub.installed() as calls:
        with mmf_amd.fp32_training():
            out = model(SampleList(sample))
        (key, loss), = out["losses"].items()
        loss.backward()
        names = {c[0] for c in calls}
    assert not (names & BF16_KERNELS), names & BF16_KERNELS
    assert all(p.grad is not None and p.grad.dtype == torch.float32 for p in model.parameters())     # the pooler is in use here


def test_in_batch_pairs_builds_an_fp32_graph():
    """ViLBERT's `in_batch_pairs` batch expansion (vilbert.py:678-710), refused in fp32 until round 5: the dry run issues the fp32 expansion kernels and
    no bf16 one."""
    z, case, cfg, sd, sample = G.load_vilbert_case("vilbert_pairs")
    model = MU.build_vilbert(cfg, sd, device="cpu")
    model.eval()
    with native_stub.installed() as calls:
        with mmf_amd.fp32_training():
            out = model(SampleList({k: v for k, v in sample.items() if k != "targets"}))
        out["scores"].sum().backward()
        names = [c[0] for c in calls]
    B = sample["input_ids"].shape[0]
    assert out["scores"].shape[0] == B * B and out["scores"].dtype == torch.float32
    assert names.count("expand_batch_f32") == 2 and names.count("reduce_batch_f32") == 2 and "expand_batch" not in names and not (set(names) & BF16_KERNELS)


def _fp32_step(model, sample, train=True):
    model.train(train)
    with native_stub.installed() as calls:
        with mmf_amd.fp32_training():
            out = model(SampleList(sample))
        loss = sum(v.sum() for v in out["losses"].values())
        loss.backward()
        names = {c[0] for c in calls}
    assert not (names & (BF16_KERNELS | {"masked_mean_fwd", "masked_mean_bwd", "rowgroup_scale", "rowgroup_scale_bwd", "align_pos_bwd", "soft_target_kl_bwd",
                                         "l2norm_rows_fwd", "l2norm_rows_bwd", "gather_rows2", "ptr_scores_fwd", "ptr_scores_bwd", "rows_scatter_add",
                                         "cast_bf16_to_f32", "cast2d_f32_to_bf16"})), names
    assert all(p.grad is None or p.grad.dtype == torch.float32 for p in model.parameters())
    return names


def test_round5_operators_build_fp32_graphs():
    """What round 5 added to mmf_amd.fp32_training(): ViLBERT's dynamic_attention gate and masked-region head,
    VisualBERT's image_text_alignment, M4C's stages — every launch an fp32 kernel, every gradient fp32."""
    z, case, cfg, sd, sample = G.load_vilbert_case("vilbert_dyn")
    model = MU.build_vilbert(cfg, sd, device="cpu")
    names = _fp32_step(model, sample)
    assert {"masked_mean_f32", "masked_mean_f32_bwd", "rowgroup_scale_f32", "rowgroup_scale_f32_bwd", "gate_sigmoid_fwd", "gate_sigmoid_bwd"} <= names
    assert all(p.grad is not None for n, p in model.named_parameters() if "dyLinear" in n)
    z, case, cfg, sd, sample = G.load_vilbert_pretraining_case(0)
    model = MU.build_vilbert_pretraining(cfg, sd, device="cpu", visual_target=0)
    names = _fp32_step(model, {k: v for k, v in sample.items() if not k.startswith("_")})
    assert {"soft_target_kl_fwd", "soft_target_kl_f32_bwd", "vocab_cross_entropy_f32_bwd"} <= names
    assert model.model.cls.imagePredictions.decoder.weight.grad is not None
    for vt, kernel in ((1, "mse_f32_bwd"), (2, "nce_f32_bwd")):      # the non-default masked-region targets
        z, case, cfg, sd, sample = G.load_vilbert_pretraining_case(vt)
        model = MU.build_vilbert_pretraining(cfg, sd, device="cpu", visual_target=vt, **(dict(num_negative=cfg["num_negative"]) if vt == 2 else {}))
        names = _fp32_step(model, {k: v for k, v in sample.items() if not k.startswith("_")})
        assert kernel in names and "mse_bwd" not in names and "nce_bwd" not in names
    z, case, cfg, sd, sample = G.load_case("align64")
    model = MU.build_visual_bert(cfg, sd, device="cpu")
    names = _fp32_step(model, sample)
    assert {"align_pos_fwd", "align_pos_f32_bwd"} <= names
    z, case, cfg, sd, sample = G.load_m4c_case()
    model = MU.build_m4c(cfg, sd, device="cpu")
    names = _fp32_step(model, sample)
    assert {"l2norm_rows_f32", "l2norm_rows_f32_bwd", "gather_rows2_f32", "ptr_scores_f32", "ptr_scores_f32_bwd", "slice_rows_f32"} <= names
    assert all(p.grad is not None and p.grad.shape == p.shape for n, p in model.named_parameters())


def test_visual_bert_pretraining_step_builds_an_fp32_graph():
    z, case, cfg, sd, sample = G.load_pretraining_case()
    model = MU.build_visual_bert_pretraining(cfg, sd, device="cpu")
    model.train()
    with native_stub.installed() as calls:
        with mmf_amd.fp32_training():
            out = model(SampleList(sample))
        (key, loss), = out["losses"].items()
        loss.backward()
        names = {c[0] for c in calls}
    assert not (names & BF16_KERNELS), names & BF16_KERNELS
    assert "vocab_cross_entropy_f32_bwd" in names
    word = model.model.bert.embeddings.word_embeddings.weight
    assert word.grad is not None and word.grad.dtype == torch.float32 and model.model.cls.predictions.decoder.weight is word


@pytest.mark.parametrize("name", ["vilbert_small", "vilbert_nlvr2"])
def test_vilbert_step_builds_an_fp32_graph_only(name):
    z, case, cfg, sd, sample = G.load_vilbert_case(name)
    over = dict(training_head_type="nlvr2", losses=[dict(type="cross_entropy")]) if name == "vilbert_nlvr2" else {}
    model = MU.build_vilbert(cfg, sd, device="cpu", **over)
    model.train()
    with native_stub.installed() as calls:
        with mmf_amd.fp32_training():
            out = model(SampleList(sample))
        (key, loss), = out["losses"].items()
        loss.backward()
        names = {c[0] for c in calls}
    assert not (names & BF16_KERNELS), names & BF16_KERNELS
    assert any(c[0] == "attention_f32_bwd" and c[3] != c[4] for c in calls)          # the co-attention backward: Sq != Sk
    missing = [k for k, p in model.named_parameters() if p.grad is None and "q_dense" not in k]
    assert not missing, missing          # (q_dense1 / q_dense2 exist in the reference's parameter tree and are never called)
    assert all(p.grad is None or p.grad.dtype == torch.float32 for p in model.parameters())


def test_mmbt_and_mmft_steps_build_fp32_graphs():
    from oracle.mmbt_oracle import SHARED
    from oracle.mmft_oracle import shared
    for load, build, extra in ((G.load_mmbt_case, lambda cfg, sd: MU.build_mmbt(cfg, sd, SHARED, device="cpu"), set()),
                               (G.load_mmft_case, lambda cfg, sd: MU.build_mmft(cfg, sd, shared(cfg), device="cpu"), set())):
        z, case, cfg, sd, sample = load()
        model = build(cfg, sd)
        model.train()
        with native_stub.installed() as calls:
            with mmf_amd.fp32_training():
                out = model(SampleList(sample))
            (key, loss), = out["losses"].items()
            loss.backward()
            names = {c[0] for c in calls}
        assert not (names & BF16_KERNELS), names & BF16_KERNELS
        assert "attention_f32_bwd" in names and "scatter_add_rows_f32" in names
        assert all(p.grad is None or p.grad.dtype == torch.float32 for p in model.parameters())
        assert sum(p.grad is not None for p in model.parameters()) > 20


def test_uniter_step_builds_an_fp32_graph():
    z, case, cfg, sd, sample = G.load_uniter_case()
    model = MU.build_uniter(cfg, sd, device="cpu")
    model.train()
    with native_stub.installed() as calls:
        with mmf_amd.fp32_training():
            out = model(SampleList(sample))
        (key, loss), = out["losses"].items()
        loss.sum().backward()
        names = {c[0] for c in calls}
    assert not (names & BF16_KERNELS), names & BF16_KERNELS
    assert all(p.grad is None or p.grad.dtype == torch.float32 for p in model.parameters())
    assert sum(p.grad is not None for p in model.parameters()) > 20
