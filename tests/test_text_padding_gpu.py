"""Text-padding trimming on the MI355X path (mmf_amd/common/prefetch.py::trim_text_padding, mmf_amd/utils/graph.py::BucketedTrainStep): the HIP model on
the trimmed batch gives the scores, loss and parameter gradients of the untrimmed batch (the reference never compacts, mmf/models/visual_bert.py:94-106;
tests/test_text_padding_cpu.py shows the identity on the pinned oracle), and the bucketed graphed step trains like the eager step on untrimmed batches."""
import pytest
import torch

from mmf_amd.common.prefetch import trim_text_padding
from mmf_amd.common.sample import SampleList
from oracle import visual_bert_oracle as O
from tests.golden_utils import load_case
from tests.model_utils import build_visual_bert, sample_to
from tests.test_text_padding_cpu import padded_sample

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(model, batch):
    model.zero_grad(set_to_none=True)
    out = model(SampleList(sample_to(dict(batch), "cuda")))
    loss = sum(v.sum() for v in out["losses"].values())
    loss.backward()
    return out["scores"].detach().float().cpu(), float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("fp32", [False, True])
def test_trimmed_batch_gives_the_untrimmed_results(fp32):
    import contextlib
    import mmf_amd
    cfg, sd, s = padded_sample((5, 3, 7), 24)
    t = trim_text_padding(SampleList(s), 8)
    assert t["input_ids"].shape == (3, 8)
    model = build_visual_bert(cfg, sd)
    model.eval()
    with (mmf_amd.fp32_training() if fp32 else contextlib.nullcontext()):
        s0, l0, g0 = run(model, s)
        s1, l1, g1 = run(model, t)
    # both runs round the same per-row arithmetic; what differs is the order of (fewer) zero terms in the softmax sums and the weight-gradient
    # reductions over fewer rows: far inside one bf16 rounding of the scores, and inside fp32 rounding on the fp32 path
    tol_s, tol_g = (1e-5, 1e-4) if fp32 else (4e-3, 1e-2)
    assert float((s0 - s1).abs().max()) <= tol_s * max(1.0, float(s0.abs().max()))
    assert abs(l0 - l1) <= tol_s * abs(l0)
    assert set(g0) == set(g1)
    for n in g0:
        if n.endswith("self.key.bias") or n.endswith("position_embeddings.weight"):
            continue
        assert rel(g1[n], g0[n]) <= tol_g, (n, rel(g1[n], g0[n]))
    pos0, pos1 = g0["model.bert.embeddings.position_embeddings.weight"], g1["model.bert.embeddings.position_embeddings.weight"]
    assert float(pos0[8:24].abs().max()) == 0.0 and float(pos1[8:].abs().max()) == 0.0      # rows no sample uses: exactly nothing, trimmed or not
    assert rel(pos1[:8], pos0[:8]) <= tol_g
    # and against the pinned oracle on the UNTRIMMED batch (north_star's bounds)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.train_step_loss(sdr, cfg, s, train=False)
    bound = 1e-3 if fp32 else 5e-2
    assert float((s1 - ref["scores"].detach()).abs().max()) <= bound


def test_bucketed_graphed_step_trains_like_the_eager_step_on_untrimmed_batches():
    """Three updates on batches of three different text lengths: BucketedTrainStep (one captured step per length bucket, shared optimizer state)
    against the eager loop on the untrimmed batches, eval mode (no dropout)."""
    from mmf_amd.modules.optimizers import AdamW
    from mmf_amd.utils.graph import BucketedTrainStep
    batches = []
    for lens in ((5, 3, 7), (12, 14, 9), (24, 2, 2), (6, 6, 6)):
        cfg, sd, s = padded_sample(lens, 24)
        batches.append(SampleList(sample_to(s, "cuda")))

    def fresh():
        m = build_visual_bert(cfg, sd)
        m.eval()
        return m, AdamW(m.parameters(), lr=1e-3, weight_decay=0.01, capturable=True)

    m1, o1 = fresh()
    step = BucketedTrainStep(m1, optimizer=o1, warmup=1, trim=8)
    losses = [float(step(b)) for b in batches]
    assert len(step.steps) == 3 and float(o1._dev_state[0]) == 4.0      # (5,3,7) and (6,6,6) share the 8-column bucket; captures update nothing
    shapes = sorted(int(g.static_batch["input_ids"].shape[1]) for g in step.steps.values())
    assert shapes == [8, 16, 24]
    m2, o2 = fresh()
    ref_losses = []
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        for b in batches:
            out = m2(b)
            loss = sum(v.sum() for v in out["losses"].values())
            m2.zero_grad(set_to_none=True)
            loss.backward()
            o2.step()
            ref_losses.append(float(loss))
            del out, loss
    torch.cuda.synchronize()
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 5e-3 * abs(b), (losses, ref_losses)
    worst = 0.0
    for (n, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        worst = max(worst, float((p - q).abs().max()))
    assert worst <= 4e-4, worst         # 4 Adam steps of 1e-3: a skipped / doubled / stale-gradient update would show at 1e-3
    # p.grad follows the bucket that ran last
    last = [g for g in step.steps.values() if int(g.static_batch["input_ids"].shape[1]) == 8][0]
    assert all(p.grad is g for p, g in zip(last.params, last.grads))


@pytest.mark.parametrize("which", ["vilbert", "mmbt"])
def test_trimmed_batch_on_the_two_stream_and_the_mmbt_models(which):
    """The same identity on the HIP path for ViLBERT (text mask in the text stream and in the image -> text co-attention) and MMBT (text at the END of the
    joint sequence); oracle side: tests/test_text_padding_cpu.py."""
    from tests.golden_utils import load_mmbt_case, load_vilbert_case
    from tests.model_utils import build_mmbt, build_vilbert
    from tests.test_text_padding_cpu import _padded
    if which == "vilbert":
        z, case, cfg, sd, sample = load_vilbert_case("vilbert_small")
        model = build_vilbert(cfg, sd)
        s = _padded(sample, (5, 3, 7), 24, cfg["vocab_size"])
    else:
        z, case, cfg, sd, sample = load_mmbt_case("mmbt_small64")
        from oracle import mmbt_oracle as OM
        model = build_mmbt(cfg, sd, OM.SHARED)
        s = _padded(sample, (5, 3, 7, 8), 24, cfg["vocab_size"])
    model.eval()
    t = trim_text_padding(SampleList(s), 8)
    assert t["input_ids"].shape[1] == 8
    s0, l0, g0 = run(model, s)
    s1, l1, g1 = run(model, t)
    assert float((s0 - s1).abs().max()) <= 4e-3 * max(1.0, float(s0.abs().max()))
    assert abs(l0 - l1) <= 4e-3 * abs(l0)
    assert set(g0) == set(g1)
    for n in g0:
        if n.endswith(("key.bias", "key1.bias", "key2.bias")) or "position_embeddings" in n:
            continue
        if float(g0[n].abs().max()) == 0.0:
            assert float(g1[n].abs().max()) == 0.0, n
            continue
        assert rel(g1[n], g0[n]) <= 2e-2, (n, rel(g1[n], g0[n]))
