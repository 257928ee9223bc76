"""Generate tests/golden/visual_bert_*.npz by running the ACTUAL reference implementation
(/root/reference/mmf/models/visual_bert.py + modules/embeddings.py + modules/hf_layers.py +
modules/losses.py, HF transformers for the un-vendored Bert blocks) in the build container.

    python tests/golden/make_golden.py

The reference tree exists only in the build container, so the outputs are committed as small
fixtures; weights are NOT stored — they are regenerated bit-exactly by detweights.state_dict().
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refshim  # noqa: E402

refshim.install()
import detweights  # noqa: E402
from omegaconf import OmegaConf  # noqa: E402  (the shim)
import mmf.models.visual_bert as ref_vb  # noqa: E402
from mmf.modules.losses import LogitBinaryCrossEntropy  # noqa: E402

CASES = {
    # head_dim 16: oracle-only pin
    "tiny": dict(hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64, vocab_size=100,
                 max_position_embeddings=24, visual_embedding_dim=16, num_labels=7, B=3, T=8, R=5, seed=11),
    # head_dim 64, ragged everything: oracle pin + HIP-path pin (tests/test_model_parity_gpu.py)
    "small64": dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=211,
                    max_position_embeddings=40, visual_embedding_dim=72, num_labels=37, B=3, T=12, R=7, seed=12),
}


class SampleList(dict):
    __getattr__ = dict.get


def reference_config(c):
    return OmegaConf.create(dict(
        bert_model_name=None, training_head_type="classification", visual_embedding_dim=c["visual_embedding_dim"],
        special_visual_initialize=True, embedding_strategy="plain", bypass_transformer=False,
        output_attentions=False, output_hidden_states=True, random_initialize=False, freeze_base=False,
        finetune_lr_multiplier=1, pooler_strategy="vqa", zerobias=False,
        hidden_size=c["hidden_size"], num_hidden_layers=c["num_hidden_layers"],
        num_attention_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
        vocab_size=c["vocab_size"], max_position_embeddings=c["max_position_embeddings"], type_vocab_size=2,
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, hidden_act="gelu", layer_norm_eps=1e-12,
        initializer_range=0.02, num_labels=c["num_labels"], losses=[dict(type="logit_bce")], model="visual_bert"))


def make_inputs(c):
    seed, B, T, R = c["seed"], c["B"], c["T"], c["R"]
    ids = (detweights.uniform(B * T, seed + 100) * c["vocab_size"]).astype(np.int64).reshape(B, T)
    mask = np.ones((B, T), dtype=np.int64)
    mask[1, T // 2:] = 0           # a padded question
    mask[2, T - 2:] = 0
    seg = (detweights.uniform(B * T, seed + 101) > 0.7).astype(np.int64).reshape(B, T)
    feats = detweights.uniform(B * R * c["visual_embedding_dim"], seed + 102).astype(np.float32).reshape(B, R, -1)
    max_features = np.array([R, R - 2, R - 1], dtype=np.int64)[:B]
    targets = np.zeros((B, c["num_labels"]), dtype=np.float32)
    for b in range(B):
        targets[b, (3 * b + 1) % c["num_labels"]] = 1.0
        targets[b, (5 * b + 2) % c["num_labels"]] = 0.6
    return dict(input_ids=ids, input_mask=mask, segment_ids=seg, image_feature_0=feats, max_features=max_features,
                targets=targets)


def main():
    for name, c in CASES.items():
        cfg = reference_config(c)
        model = ref_vb.VisualBERT(cfg)
        model.build()
        model.eval()  # dropout off: parity is defined in eval mode (SURVEY.md §7 hard parts)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("position_ids")}
        sd = detweights.state_dict(shapes, c["seed"])
        missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not unexpected and all(k.endswith("position_ids") for k in missing), (missing, unexpected)
        inp = make_inputs(c)
        sl = SampleList(
            input_ids=torch.from_numpy(inp["input_ids"]), input_mask=torch.from_numpy(inp["input_mask"]),
            segment_ids=torch.from_numpy(inp["segment_ids"]), image_feature_0=torch.from_numpy(inp["image_feature_0"]),
            image_info_0=SampleList(max_features=torch.from_numpy(inp["max_features"])),
            targets=torch.from_numpy(inp["targets"]), dataset_name="vqa2", dataset_type="train")
        out = model.forward(sl)
        loss = LogitBinaryCrossEntropy()(sl, out)
        loss.backward()
        rec = {"in_" + k: v for k, v in inp.items()}
        rec["scores"] = out["scores"].detach().numpy()
        rec["sequence_output"] = out["sequence_output"].detach().numpy()
        rec["pooled_output"] = out["pooled_output"].detach().numpy()
        rec["loss"] = np.array(loss.item(), dtype=np.float64)
        names, norms, sums = [], [], []
        for k, p in model.named_parameters():
            g = p.grad
            names.append(k)
            norms.append(0.0 if g is None else float(g.double().norm()))
            sums.append(0.0 if g is None else float(g.double().sum()))
            # full gradients of the small tensors (biases, LayerNorms, type tables)
            if g is not None and g.numel() <= 4096:
                rec["grad::" + k] = g.numpy()
        rec["grad_names"] = np.array(names)
        rec["grad_norms"] = np.array(norms)
        rec["grad_sums"] = np.array(sums)
        rec["param_names"] = np.array(list(shapes.keys()))
        rec["param_shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
        rec["case"] = np.array(repr(c))
        path = os.path.join(HERE, "visual_bert_%s.npz" % name)
        np.savez_compressed(path, **rec)
        print(name, "loss", loss.item(), "scores[0,:4]", rec["scores"][0, :4], "->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
