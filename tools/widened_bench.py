"""Training-step throughput of the widened rows at the shapes BASELINE.json's other configs
name, on one MI355X: forward + loss + backward + fused AdamW, train mode (dropout on), synthetic inputs resident in HBM,
random-init weights.  Steps are launched eagerly (no hipGraph: several of these models branch on tensor values in their
input massaging, as the reference does), so the small ones are partly host-bound — `gpu_busy_ms` (sum of kernel time from
HIP events around one step's GEMMs is not enough for that) is therefore reported as the GEMM time only, next to wall time.

    python tools/widened_bench.py [mmbt vilbert uniter mmft m4c m4c:graph]

Prints one line per model and writes gpurun_out/widened_bench.json."""
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mmf_amd  # noqa: E402,F401
from bench import KernelProbe  # noqa: E402
from mmf_amd.common.registry import registry  # noqa: E402
from mmf_amd.common.sample import SampleList  # noqa: E402
from mmf_amd.utils.configuration import Config  # noqa: E402
from tests import model_utils as MU  # noqa: E402  (config builders only)

DEV = "cuda"
BERT = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
            max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12)
LARGE = dict(BERT, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)


def text(B, T, g):
    ids = torch.randint(1000, 30522, (B, T), generator=g)
    ids[:, 0] = 101
    return dict(input_ids=ids, input_mask=torch.ones(B, T, dtype=torch.long), segment_ids=torch.zeros(B, T, dtype=torch.long))


def soft_targets(B, n, g):
    t = torch.zeros(B, n)
    for b in range(B):
        t[b, torch.randperm(n, generator=g)[:3]] = torch.tensor([1.0, 0.6, 0.3])
    return t


def case_mmbt(g):
    B = 8
    cfg = dict(BERT, modal_hidden_size=2048, num_labels=2)
    s = dict(text(B, 128, g), image_feature_0=torch.rand(B, 100, 2048, generator=g), targets=torch.randint(0, 2, (B,), generator=g),
             dataset_name="hateful_memes", dataset_type="train")
    return "MMBT Hateful Memes (100x2048 features + 128 tok, BERT-base)", B, MU.build_mmbt(cfg), s


def case_vilbert(g):
    B = 32
    cfg = dict(BERT, pad_token_id=0, v_feature_size=2048, v_hidden_size=1024, v_num_hidden_layers=6, v_num_attention_heads=8,
               v_intermediate_size=1024, bi_hidden_size=1024, bi_num_attention_heads=8, v_biattention_id=[0, 1, 2, 3, 4, 5],
               t_biattention_id=[6, 7, 8, 9, 10, 11], fusion_method="mul", num_labels=3129)
    s = dict(text(B, 128, g), image_feature_0=torch.rand(B, 100, 2048, generator=g),
             image_info_0={"max_features": torch.full((B,), 100, dtype=torch.long), "bbox": torch.rand(B, 100, 5, generator=g)},
             targets=soft_targets(B, 3129, g), dataset_name="vqa2", dataset_type="train")
    return "ViLBERT VQA2 (two streams, 6 co-attention layers, 100 regions + 128 tok)", B, MU.build_vilbert(cfg), s


def _boxes(B, R, g):
    xy = torch.rand(B, R, 4, generator=g) * 0.45
    return torch.stack([xy[..., 0], xy[..., 1], xy[..., 0] + xy[..., 2] + 0.05, xy[..., 1] + xy[..., 3] + 0.05], dim=-1)


def case_uniter(g):
    B = 64
    cfg = dict(LARGE, img_dim=2048, head_hidden_size=2048, num_labels=3129)
    s = dict(text(B, 128, g), image_feature_0=torch.rand(B, 100, 2048, generator=g),
             image_info_0={"max_features": torch.full((B,), 100, dtype=torch.long), "bbox": _boxes(B, 100, g),
                           "image_width": torch.full((B,), 640), "image_height": torch.full((B,), 480)},
             targets=soft_targets(B, 3129, g), dataset_name="vqa2", dataset_type="train")
    return "UNITER 24-layer H=1024 joint encoder (S = 228)", B, MU.build_uniter(cfg), s


def case_mmft(g):
    B = 64
    H = LARGE["hidden_size"]
    cfg = dict(LARGE, num_labels=2, modalities=[
        dict(type="text", key="text", position_dim=512, segment_id=0, embedding_dim=H, layer_norm_eps=1e-12, hidden_dropout_prob=0.1),
        dict(type="image", key="image", embedding_dim=2048, position_dim=512, segment_id=1, layer_norm_eps=1e-12, hidden_dropout_prob=0.1)])
    s = dict(text(B, 128, g), image=torch.rand(B, 100, 2048, generator=g), image_mask=torch.ones(B, 100, dtype=torch.long),
             targets=torch.randint(0, 2, (B,), generator=g), dataset_name="hateful_memes", dataset_type="train")
    return "MMF Transformer 24-layer H=1024 (text 128 + image 100x2048)", B, MU.build_mmft(cfg), s


def case_m4c(g):
    B = 32
    registry.register("config", Config({"datasets": "textvqa"}))
    registry.register("textvqa_num_final_outputs", 5050)
    registry.register("textvqa_answer_processor", Config({"BOS_IDX": 1}))
    model = registry.get_model_class("m4c")(Config({"model": "m4c", "text_bert_init_from_bert_base": False}))
    model.build(); model.init_losses()
    s = {"text": torch.randint(1000, 30522, (B, 20), generator=g), "text_len": torch.randint(5, 21, (B,), generator=g),
         "image_feature_0": torch.rand(B, 100, 2048, generator=g), "obj_bbox_coordinates": torch.rand(B, 100, 4, generator=g),
         "image_info_0": {"max_features": torch.full((B,), 100, dtype=torch.long)},
         "context_feature_0": torch.randn(B, 50, 300, generator=g), "context_feature_1": torch.rand(B, 50, 604, generator=g),
         "image_feature_1": torch.rand(B, 100, 2048, generator=g), "ocr_bbox_coordinates": torch.rand(B, 50, 4, generator=g),
         "context_info_0": {"max_features": torch.randint(5, 51, (B,), generator=g)}, "order_vectors": torch.zeros(B, 50, 50),
         "train_prev_inds": torch.randint(0, 5050, (B, 12), generator=g), "targets": (torch.rand(B, 12, 5050, generator=g) > 0.999).float(),
         "train_loss_mask": (torch.rand(B, 12, generator=g) > 0.3).float(), "dataset_name": "textvqa", "dataset_type": "train"}
    return "M4C TextVQA (20 + 100 + 50 + 12 positions, 3-layer text BERT + 4-layer MMT, 5000 + 50 scores)", B, model.to(DEV), s


CASES = dict(mmbt=case_mmbt, vilbert=case_vilbert, uniter=case_uniter, mmft=case_mmft, m4c=case_m4c)


def to_dev(s):
    out = {}
    for k, v in s.items():
        out[k] = v.to(DEV) if isinstance(v, torch.Tensor) else (to_dev(v) if isinstance(v, dict) else v)
    return out


def run(name, steps=10, warmup=3):
    name, _, mode = name.partition(":")        # "m4c:graph" replays the whole update from one hipGraph
    g = torch.Generator().manual_seed(1234)
    torch.manual_seed(1234)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        label, B, model, sample = CASES[name](g)
    model = model.to(DEV).train()
    batch = SampleList(to_dev(sample))
    full = Config(model=name, optimizer=dict(params=dict(lr=5e-5)), model_config={name: model.config})
    opt = registry.get_optimizer_class("adam_w")(model.get_optimizer_parameters(full), lr=5e-5, eps=1e-8, capturable=(mode == "graph"))

    def step():
        model.zero_grad(set_to_none=True)
        out = model(batch)
        loss = sum(v.sum() for v in out["losses"].values())
        loss.backward()
        opt.step()
        return loss

    eager_step = step
    if mode == "graph":
        from mmf_amd.utils.graph import GraphedTrainStep
        step = GraphedTrainStep(model, batch, warmup=2, optimizer=opt)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    if mode == "graph":       # the step is one replayed hipGraph: no per-GEMM events
        gemm_ms = gemm_fl = None
    else:
        with KernelProbe() as probe:
            eager_step()
        by = {k: d for k, d in probe.summary().items() if k.startswith("gemm")}
        gemm_ms = sum(d["ms"] for d in by.values()); gemm_fl = sum(d["work"] for d in by.values())
    rec = dict(model=name, workload=label, batch=B, ms_per_step=round(ms, 3), samples_per_s=round(B / ms * 1e3, 1),
               loss=round(float(last.item()), 4), params=sum(p.numel() for p in model.parameters()),
               gemm_flop_per_step=gemm_fl, gemm_ms_per_step=None if gemm_ms is None else round(gemm_ms, 3),
               gemm_tflops=None if gemm_ms is None else round(gemm_fl / gemm_ms / 1e9, 1),
               step_tflops=None if gemm_fl is None else round(gemm_fl / ms / 1e9, 1), launch=mode or "eager", dtype="bf16", data="synthetic")
    detail = "(one hipGraph: no per-GEMM events)" if gemm_ms is None else "GEMMs %.1f GFLOP in %.2f ms (%.0f TFLOP/s), whole step %.0f TFLOP/s" % (
        gemm_fl / 1e9, gemm_ms, rec["gemm_tflops"], rec["step_tflops"])
    print("%-8s B=%3d  %8.2f ms/step  %8.1f samples/s  %s  loss %.4f" % (name, B, ms, rec["samples_per_s"], detail, rec["loss"]), flush=True)
    del model, opt, batch
    torch.cuda.empty_cache()
    return rec


if __name__ == "__main__":
    which = sys.argv[1:] or list(CASES)
    recs = []
    for n in which:
        try:
            recs.append(run(n))
        except Exception as e:      # noqa: BLE001  (keep going: one line per model is the point)
            import traceback
            traceback.print_exc()
            recs.append(dict(model=n, error=repr(e)))
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(recs, open(os.path.join(out, "widened_bench%s.json" % ("_" + "_".join(n.replace(":", "-") for n in which) if any(":" in n for n in which) else "")), "w"), indent=1)
