"""Host-side dry run of one training step (train mode, dropout on, fused AdamW) of every registered model on CPU.  The kernel
wrappers are replaced by extent / dtype / leading-dimension checkers (tests/native_stub.py), so this exercises the Python half
of each path — every autograd Function's forward and backward, buffer sizes, argument order, the optimizer's parameter
groups — and compares WHICH parameters receive a gradient with the reference (the fixtures record the reference's gradient
norms: a parameter the reference leaves without gradient must stay without one here, and vice versa).  The numbers are the
`-m gpu` tests' job."""
import contextlib

import pytest
import torch

from mmf_amd.common.registry import registry
from mmf_amd.common.sample import SampleList
from mmf_amd.utils.configuration import Config
from tests import golden_utils as G
from tests import model_utils as MU
from tests import native_stub


def _visual_bert():
    z, case, cfg, sd, sample = G.load_case("small64")
    return z, MU.build_visual_bert(cfg, sd, device="cpu"), sample, "model."


def _nlvr2():
    z, case, cfg, sd, sample = G.load_nlvr2_case()
    return z, MU.build_visual_bert(cfg, sd, device="cpu", training_head_type="nlvr2", pooler_strategy="default",
                                   losses=[dict(type="cross_entropy")]), sample, "model."


def _bypass():
    z, case, cfg, sd, sample = G.load_bypass_case()
    return z, MU.build_visual_bert(cfg, sd, device="cpu", bypass_transformer=True, pooler_strategy="default"), sample, "model."


def _pretraining():
    z, case, cfg, sd, sample = G.load_pretraining_case()
    return z, MU.build_visual_bert_pretraining(cfg, sd, device="cpu"), sample, "model."


def _mmbt():
    z, case, cfg, sd, sample = G.load_mmbt_case()
    from oracle.mmbt_oracle import SHARED
    return z, MU.build_mmbt(cfg, sd, SHARED, device="cpu"), sample, "model."


def _mmbt_pretraining():
    z, case, cfg, sd, sample = G.load_mmbt_pretraining_case()
    from oracle.mmbt_oracle import SHARED
    return z, MU.build_mmbt_pretraining(cfg, sd, SHARED, device="cpu"), sample, "model."


def _mmft():
    z, case, cfg, sd, sample = G.load_mmft_case()
    from oracle.mmft_oracle import shared
    return z, MU.build_mmft(cfg, sd, shared(cfg), device="cpu"), sample, ""


def _vilbert():
    z, case, cfg, sd, sample = G.load_vilbert_case()
    return z, MU.build_vilbert(cfg, sd, device="cpu"), sample, "model."


def _vilbert_pairs():
    z, case, cfg, sd, sample = G.load_vilbert_case("vilbert_pairs")
    return z, MU.build_vilbert(cfg, sd, device="cpu"), sample, "model."


def _vilbert_fast():
    z, case, cfg, sd, sample = G.load_vilbert_case("vilbert_fast")
    return z, MU.build_vilbert(cfg, sd, device="cpu"), sample, "model."


def _vilbert_pretraining():
    z, case, cfg, sd, sample = G.load_vilbert_pretraining_case()
    return z, MU.build_vilbert_pretraining(cfg, sd, device="cpu"), sample, "model."


def _uniter():
    z, case, cfg, sd, sample = G.load_uniter_case()
    return z, MU.build_uniter(cfg, sd, device="cpu"), sample, ""


def _m4c():
    z, case, cfg, sd, sample = G.load_m4c_case()
    return z, MU.build_m4c(cfg, sd, device="cpu"), sample, ""


CASES = {"visual_bert": _visual_bert, "visual_bert_nlvr2": _nlvr2, "visual_bert_pretraining": _pretraining, "visual_bert_bypass": _bypass, "mmbt": _mmbt, "mmbt_pretraining": _mmbt_pretraining, "mmft": _mmft, "vilbert": _vilbert, "vilbert_pairs": _vilbert_pairs, "vilbert_fast": _vilbert_fast, "vilbert_pretraining": _vilbert_pretraining, "uniter": _uniter,
         "m4c": _m4c}


@pytest.mark.parametrize("name", sorted(CASES))
def test_training_step_plumbing(name):
    z, model, sample, prefix = CASES[name]()
    model.train()
    key = name.split("_nlvr2")[0].split("_pretraining")[0].split("_bypass")[0].split("_pairs")[0].split("_fast")[0]
    full = Config(model=key, optimizer=dict(params=dict(lr=5e-5)), model_config={key: model.config})
    opt = registry.get_optimizer_class("adam_w")(model.get_optimizer_parameters(full), lr=5e-5, eps=1e-8)
    with native_stub.installed() as calls:
        if name == "vilbert_pairs":                  # B^2 target rows do not pass SampleList's equal-batch check (the reference's neither): loss applied here
            out = model(SampleList({k: v for k, v in sample.items() if k != "targets"}))
            out["losses"] = {"train/golden/logit_bce": torch.ops.mmf_amd.logit_bce(out["scores"], sample["targets"])}
        elif name == "vilbert_fast":                 # one text against B images (vilbert.py:712-723): ViLBERTForClassification.forward directly
            from oracle.vilbert_oracle import prepare_inputs
            p = prepare_inputs(dict(sample))
            out = model.model(p["input_ids"], p["image_feature"], p["image_location"], p["token_type_ids"], p["attention_mask"], p["image_attention_mask"])
            assert out["scores"].shape[0] == p["image_feature"].shape[0] and p["input_ids"].shape[0] == 1
            out["losses"] = {"train/golden/logit_bce": torch.ops.mmf_amd.logit_bce(out["scores"], sample["targets"])}
        else:
            out = model(SampleList(sample))
        if name == "vilbert_pretraining":            # two losses, no scores in the output (vilbert.py:1459-1469)
            assert sorted(out["losses"]) == ["coco/train/masked_img_loss", "coco/train/masked_lm_loss"]
            loss = sum(v.sum() for v in out["losses"].values())
        else:
            head = out["logits"] if name.endswith("_pretraining") else out["scores"]        # the pretraining heads return `logits`
            assert head.dtype == torch.float32 and head.shape[0] > 0
            if name == "vilbert_pairs":              # in_batch_pairs: one score row per (text, image) pair (vilbert.py:678-710)
                B = sample["input_ids"].shape[0]
                assert head.shape[0] == B * B
            assert len(out["losses"]) == 1
            (lkey, loss), = out["losses"].items()
            assert (lkey.startswith("train/") or lkey.endswith("/train/masked_lm_loss")) and loss.numel() == 1
        loss.sum().backward()
        opt.step()
    assert any(c[0] == "gemm" for c in calls) and any(c[0] == "attention_bwd" for c in calls) and any(c[0] == "adamw_multi" for c in calls)
    if name.startswith("visual_bert"):               # round 5's fused launches of the embedding stage are the ones that run
        names = [c[0] for c in calls]
        H = model.config.hidden_size
        fused_ln = H % 256 == 0 and H <= 1024        # (the fixtures are 128 wide: LayerNorm + dropout stay two launches there, as mmf_layernorm_dropout_fusable says)
        assert names.count("embed_tables_bwd") == 1          # the four small table gradients in one call (only the word table is left to rows_scatter_add)
        assert ("layernorm_dropout_fwd" in names) == fused_ln and ("layernorm_bwd_din" in names) == fused_ln
        assert sum(n == "adamw_multi" for n in names) == 1 and len(opt.param_groups) >= 2       # ONE call for all parameter groups (lr / weight decay travel per tensor)
    if name == "vilbert_pairs":                      # image rows broadcast over the text index (mode 0), text rows over the image index (mode 1)
        assert sorted(c[4] for c in calls if c[0] == "expand_batch") == [0, 1] and sorted(c[4] for c in calls if c[0] == "reduce_batch") == [0, 1]
    if name == "vilbert_fast":                       # the single text broadcast over the image batch
        assert [c[1:] for c in calls if c[0] == "expand_batch"] == [(1, 3, 12 * 128, 0)] and [c[1:] for c in calls if c[0] == "reduce_batch"] == [(1, 3, 12 * 128, 0)]
    params = dict(model.named_parameters())
    # every parameter in exactly one optimizer group
    seen = [id(p) for g in opt.param_groups for p in g["params"]]
    assert len(seen) == len(set(seen)) and set(seen) == {id(p) for p in params.values()}
    ref = {str(n): float(v) for n, v in zip(z["grad_names"], z["grad_norms"])}
    assert ref, "fixture without gradient records"
    for n, norm in ref.items():
        n = n if n in params else (n[len("model."):] if n.startswith("model.") and n[len("model."):] in params else n)
        assert n in params, n
        p = params[n]
        if norm == 0.0:
            assert p.grad is None, "%s: the reference leaves this parameter without a gradient" % n
        else:
            assert p.grad is not None and p.grad.shape == p.shape and p.grad.dtype == torch.float32, n


def test_vilbert_nce_negative_sampling_replays_the_reference_draws():
    """`visual_target: 2` (vilbert.py:1158-1203): with Tensor.random_ replaying the draws the reference run recorded, the model's sampling
    gives the same flat negative indices (70 % from other samples, 30 % from other regions of the same image), and the step runs through
    the NCE node with them."""
    z, case, cfg, sd, sample = G.load_vilbert_pretraining_case(2)
    model = MU.build_vilbert_pretraining(cfg, sd, device="cpu", visual_target=2, num_negative=cfg["num_negative"])
    B, R = z["in_image_labels"].shape
    with G.recorded_random(z):
        neg = model.model.negative_index(B, R, sample["input_ids"])
    assert torch.equal(neg, torch.from_numpy(z["in_negative_index"]))
    own = torch.arange(B).view(B, 1, 1)
    n_across = int(cfg["num_negative"] * 0.7)
    assert bool((neg[..., :n_across] // R != own).all()) and bool((neg[..., n_across:] // R == own).all())
    assert bool((neg[..., n_across:] % R != torch.arange(R).view(1, R, 1)).all())
    model.train()
    batch = {k: v for k, v in sample.items() if not k.startswith("_")}
    with native_stub.installed() as calls, G.recorded_random(z):
        out = model(SampleList(batch))
        sum(v.sum() for v in out["losses"].values()).backward()
    names = {c[0] for c in calls}
    assert "nce_fwd" in names and "nce_bwd" in names
    head = model.model.cls.imagePredictions.decoder
    assert head.weight.grad is not None and head.weight.grad.shape == head.weight.shape


def _step(model, sample):
    model.train()
    with native_stub.installed():
        out = model(SampleList(sample))
        loss = sum(v.sum() for v in out["losses"].values())
        loss.backward()
    return out


def test_visual_bert_config_branches_plumbing():
    """`pooler_strategy: default` (BertPooler on token 0), `zerobias`, `freeze_base` (visual_bert.py:121-131,168-170,389-398)."""
    z, case, cfg, sd, sample = G.load_case("small64")
    m = MU.build_visual_bert(cfg, sd, device="cpu", pooler_strategy="default")
    _step(m, sample)
    assert m.model.bert.pooler.dense.weight.grad is not None          # the pooler is on the path now
    m = MU.build_visual_bert(cfg, None, device="cpu", zerobias=True, biasfill=-2.0)      # the reference reads config.biasfill too (:347)
    assert bool((m.model.classifier[1].bias == -2.0).all())
    m = MU.build_visual_bert(cfg, sd, device="cpu", freeze_base=True)
    _step(m, sample)
    assert all(p.grad is None for n, p in m.named_parameters() if n.startswith("model.bert."))
    assert m.model.classifier[1].weight.grad is not None


def test_visual_bert_output_attentions_is_accepted_and_empty_like_the_reference():
    """`output_attentions: true` builds and runs; the `attention_weights` entry is what the reference produces: VisualBERTBase.forward calls
    `self.encoder(embedding_output, extended_attention_mask)` without `output_attentions`, so `encoded_layers[1:]` is empty
    (visual_bert.py:143-157, hf_layers.py:322-356)."""
    z, case, cfg, sd, sample = G.load_case("small64")
    m = MU.build_visual_bert(cfg, sd, device="cpu", output_attentions=True)
    m.eval()
    with native_stub.installed():
        out = m(SampleList({k: v for k, v in sample.items() if k != "targets"}))
    assert "attention_weights" in out and len(out["attention_weights"]) == 0
    m2 = MU.build_visual_bert(cfg, sd, device="cpu")
    m2.eval()
    with native_stub.installed():
        assert "attention_weights" not in m2(SampleList({k: v for k, v in sample.items() if k != "targets"}))


def test_mmbt_extras_are_accepted_and_empty_like_the_reference():
    """`output_attentions` / `output_hidden_states` of MMBT's text encoder: the reference hands `module_output[2:]` on as `extras` (mmbt.py:486-490, 545-547),
    and that tail is `(encoder_outputs[1:],)` of an encoder called WITHOUT the two arguments (mmbt.py:302-316, hf_layers.py:317-356): one empty tuple."""
    z, case, cfg, sd, sample = G.load_mmbt_case()
    from oracle.mmbt_oracle import SHARED
    conf = MU.mmbt_model_config(cfg)
    conf["text_encoder"]["params"]["output_attentions"] = True
    from mmf_amd.utils.build import build_model
    m = build_model(conf).eval()
    with native_stub.installed():
        out = m(SampleList({k: v for k, v in sample.items() if k != "targets"}))
    assert "extras" in out and len(out["extras"]) == 1 and len(out["extras"][0]) == 0
    m2 = MU.build_mmbt(cfg, sd, SHARED, device="cpu").eval()
    with native_stub.installed():
        assert "extras" not in m2(SampleList({k: v for k, v in sample.items() if k != "targets"}))


def test_mmbt_config_branches_plumbing():
    """No modal start / end tokens, `fused_feature_only`, frozen text / modal halves (mmbt.py:173-178,229-231,253-259)."""
    from oracle.mmbt_oracle import SHARED
    z, case, cfg, sd, sample = G.load_mmbt_case()
    for kw in (dict(use_modal_start_token=False, use_modal_end_token=False), dict(use_modal_start_token=True, use_modal_end_token=False)):
        m = MU.build_mmbt(dict(cfg, **kw), sd, SHARED, device="cpu")
        out = _step(m, sample)
        assert out["scores"].shape == (case["B"], cfg["num_labels"])
    m = MU.build_mmbt(cfg, sd, SHARED, device="cpu", freeze_text=True)
    _step(m, sample)
    assert m.model.bert.mmbt.transformer.encoder.layer[0].attention.self.query.weight.grad is None
    assert m.model.classifier[1].weight.grad is not None


def test_vilbert_sum_fusion_and_frozen_base_plumbing():
    """`fusion_method: sum` (vilbert.py:1315-1320) and `freeze_base` (:1429-1431)."""
    z, case, cfg, sd, sample = G.load_vilbert_case()
    m = MU.build_vilbert(dict(cfg, fusion_method="sum"), sd, device="cpu")
    _step(m, sample)
    m = MU.build_vilbert(cfg, sd, device="cpu", freeze_base=True)
    _step(m, sample)
    assert all(p.grad is None for n, p in m.named_parameters() if n.startswith("model.bert."))


@pytest.mark.parametrize("name", ["visual_bert", "mmbt", "vilbert", "uniter", "m4c"])
def test_a_cloned_static_batch_runs_through_the_model_twice(name):
    """What GraphedTrainStep does with its batch (mmf_amd/utils/graph.py::_clone_batch: a SampleList filled by item assignment, no tensor field
    recorded) and what the model makes of it: `to_device` rebuilds it (sample.py:400-420) and the forward works on a copy, so calling the model
    again on the SAME static batch sees the caller's fields only (MMBT shifts `input_ids` in its forward, VisualBERT attaches masks)."""
    from mmf_amd.utils.graph import _clone_batch
    z, model, sample, prefix = CASES[name]()
    model.train()
    static = _clone_batch(SampleList(sample))
    assert static._get_tensor_field() is None
    before = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in static.items()}
    with native_stub.installed():
        for _ in range(2):
            out = model(static)
            assert "losses" in out and len(out["losses"]) >= 1
            assert list(static.keys()) == list(before.keys())
            for k, v in before.items():
                if isinstance(v, torch.Tensor):
                    assert static[k] is not None and torch.equal(static[k], v), k


def test_deferred_weight_gradients_plumbing(monkeypatch):
    """functional.wgrad_defer (round 4): with the deferral on (what the graphed steps do) ViLBERT's connection layers hand their weight gradients to
    grouped launches of at most eight — the same set of GEMM problems as the one-by-one launches, nothing left queued when the block ends, and every
    parameter still receives a gradient tensor of its own shape."""
    from mmf_amd import functional as Fn
    from mmf_amd import _native as N
    z, model, sample, prefix = CASES["vilbert"]()
    model.train()
    seen = {}
    monkeypatch.setattr(Fn, "_WGRAD_DEFER_MIN_ROWS", 1)       # (the fixture has fewer token rows than the production threshold)
    for deferred in (False, True):
        model.zero_grad(set_to_none=True)
        with native_stub.installed() as calls:
            groups = []
            grouped = N.gemm_grouped

            def counting(problems, _g=grouped):
                groups.append(len(problems))
                return _g(problems)
            N.gemm_grouped = counting
            out = model(SampleList(sample))
            (lkey, loss), = out["losses"].items()
            ctx = Fn.wgrad_defer() if deferred else contextlib.nullcontext()
            with ctx:
                loss.sum().backward()
                if deferred:
                    assert Fn.wgrad_defer.active
            assert not Fn.wgrad_defer.active and not Fn.wgrad_defer.queues
            seen[deferred] = (sorted(c for c in calls if c[0] == "gemm"), list(groups))
        for n, p in model.named_parameters():
            assert p.grad is None or p.grad.shape == p.shape, n
    plain, deferred = seen[False], seen[True]
    assert plain[0] == deferred[0]                                   # the same GEMM problems either way
    assert sum(deferred[1]) > sum(plain[1]) and max(deferred[1]) <= N.GEMM_GROUP_MAX and len(deferred[1]) > len(plain[1])


@pytest.mark.parametrize("name", ["m4c", "visual_bert_pretraining", "mmbt_pretraining", "vilbert_pretraining", "uniter", "mmft"])
def test_shareable_weights_are_never_deferred(name, monkeypatch):
    """autograd sums the gradient contributions of a shared parameter the moment the second one arrives — before any
    flush — so a node whose weight can be tied to another node's must not hand back an unfilled dW under `wgrad_defer`: M4C's classifier
    weight is the GEMM weight of the scores node AND the lookup table of `PrevPredEmbeddings` (m4c.py:111, 361), the masked-LM decoder is tied
    to the word embeddings (visual_bert.py:179-184).  With the deferral on, every queued problem belongs to an encoder-internal matrix: no
    queued dW has the shape of a parameter that receives a second contribution, and every multiply-contributed parameter ends with a gradient."""
    from mmf_amd import functional as Fn
    z, model, sample, prefix = CASES[name]()
    model.train()
    monkeypatch.setattr(Fn, "_WGRAD_DEFER_MIN_ROWS", 1)
    pushed = []
    push = Fn.wgrad_defer.push

    def recording(prob, keep):
        pushed.append((prob["M"], prob["N"]))
        return push(prob, keep)
    monkeypatch.setattr(Fn.wgrad_defer, "push", recording)
    with native_stub.installed():
        out = model(SampleList(sample))
        loss = sum(v.sum() for v in out["losses"].values())
        with Fn.wgrad_defer():
            loss.backward()
        assert not Fn.wgrad_defer.queues and not Fn.wgrad_defer.seen
    params = dict(model.named_parameters())
    tables = {tuple(p.shape) for n, p in params.items() if "embeddings" in n or n.endswith("classifier.module.weight") or "decoder" in n or "predictions" in n}
    assert not (set(pushed) & tables), (set(pushed) & tables)
    if name == "m4c":
        w = params["classifier.module.weight"]
        assert tuple(w.shape) not in pushed and w.grad is not None and w.grad.shape == w.shape


def test_a_layer_applied_twice_is_not_queued_twice(monkeypatch):
    """Tied layers (one FeedForward node applied twice in a forward): the second weight gradient of the same matrix flushes what is queued and is
    computed immediately — the sum autograd forms right after the second node returns then reads two filled buffers, in stream order."""
    from mmf_amd import functional as Fn
    from mmf_amd import _native as N
    monkeypatch.setattr(Fn, "_WGRAD_DEFER_MIN_ROWS", 1)
    H, I, M = 64, 128, 16
    w1 = torch.nn.Parameter(torch.randn(I, H)); b1 = torch.nn.Parameter(torch.zeros(I))
    w2 = torch.nn.Parameter(torch.randn(H, I)); b2 = torch.nn.Parameter(torch.zeros(H))
    g = torch.nn.Parameter(torch.ones(H)); be = torch.nn.Parameter(torch.zeros(H))
    x = torch.randn(2, M // 2, H, requires_grad=True)
    with native_stub.installed() as calls:
        groups = []
        grouped = N.gemm_grouped

        def counting(problems, _g=grouped):
            groups.append((len(calls), len(problems)))
            return _g(problems)
        N.gemm_grouped = counting
        ff = lambda t: Fn.FeedForwardFn.apply(t, w1, b1, w2, b2, g, be, Fn.shadows.get(w1), Fn.shadows.get(w2), 1e-12, N.NO_DROP)
        y = ff(ff(x))
        with Fn.wgrad_defer():
            y.float().sum().backward()
            # the second application came first in backward and queued its two problems; the first application (same weights) found them
            # queued, launched them, and ran its own two weight gradients as plain GEMMs
            assert groups and groups[0][1] == 2
            wg = [c for c in calls if c[0] == "gemm"]
        assert not Fn.wgrad_defer.queues
    assert w1.grad is not None and w1.grad.shape == w1.shape and w2.grad is not None


def test_encoder_hands_a_materialised_per_query_mask_to_the_attention_kernels():
    """A [B, 1, S, S] additive attention mask (hf_layers.py:187-190 adds any broadcastable mask to the scores; m4c.py:424-440 builds one) reaches the
    attention launches as a [B, S, S] tensor (mmf_attn_desc.mask_query_stride), forward and backward; a [B, heads, S, S] mask goes on as one mask per head (mmf_attn_desc.mask_head_stride)."""
    from transformers import BertConfig
    from mmf_amd.modules.hf_layers import BertEncoderJit
    B, S, H = 2, 40, 128
    enc = BertEncoderJit(BertConfig(hidden_size=H, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2)).train()
    x = torch.randn(B, S, H, requires_grad=True)
    with native_stub.installed() as calls:
        out = enc(x, torch.zeros(B, 1, S, S))[0]
        out.float().sum().backward()
        att = [c for c in calls if c[0] in ("attention_fwd", "attention_bwd")]
        assert len(att) == 4 and all(c[-1] == "per-query mask" for c in att), att
        del calls[:]
        out = enc(x, torch.zeros(B, 1, 1, S))[0]
        assert [c[-1] for c in calls if c[0] == "attention_fwd"] == [0, 0]          # the key-mask form: (…, causal_tail = 0), no per-query marker
        del calls[:]
        out = enc(x, torch.zeros(B, 2, S, S))[0]
        out.float().sum().backward()
        att = [c for c in calls if c[0] in ("attention_fwd", "attention_bwd")]
        assert len(att) == 4 and all(c[-1] == "per-head mask" for c in att), att
        with pytest.raises((ValueError, RuntimeError)):
            enc(x, torch.zeros(B, 2, S, S + 1))
