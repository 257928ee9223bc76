"""Generate tests/golden/visual_bert_bypass.npz: the reference VisualBERT with `bypass_transformer: true` (mmf/models/visual_bert.py:
52-56, 116-141 — the text goes through the 12-layer encoder alone, the visual embeddings join it only in one `additional_layer`) run by
the reference's own code, like make_golden.py::main does for the plain model.

    python tests/golden/make_visual_bert_bypass.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG  # noqa: E402

detweights, SampleList, ref_vb = MG.detweights, MG.SampleList, MG.ref_vb


def main():
    c = dict(MG.CASES["small64"], seed=63)
    cfg = MG.reference_config(c)
    cfg["bypass_transformer"] = True
    cfg["output_hidden_states"] = False          # (asserted off on the bypass path, visual_bert.py:121-123)
    cfg["pooler_strategy"] = "default"           # the bypass path's pooled output feeds the head
    model = ref_vb.VisualBERT(cfg)
    model.build()
    model.eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("position_ids")}
    sd = detweights.state_dict(shapes, c["seed"])
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("position_ids") for k in missing), (missing, unexpected)
    inp = MG.make_inputs(c)
    sl = SampleList(
        input_ids=torch.from_numpy(inp["input_ids"]), input_mask=torch.from_numpy(inp["input_mask"]),
        segment_ids=torch.from_numpy(inp["segment_ids"]), image_feature_0=torch.from_numpy(inp["image_feature_0"]),
        image_info_0=SampleList(max_features=torch.from_numpy(inp["max_features"])),
        targets=torch.from_numpy(inp["targets"]), dataset_name="vqa2", dataset_type="train")
    out = model.forward(sl)
    loss = MG.LogitBinaryCrossEntropy()(sl, out)
    loss.backward()
    rec = {"in_" + k: v for k, v in inp.items()}
    rec["scores"] = out["scores"].detach().numpy()
    rec["loss"] = np.array(loss.item(), dtype=np.float64)
    names, norms = [], []
    for k, p in model.named_parameters():
        g = p.grad
        names.append(k)
        norms.append(0.0 if g is None else float(g.double().norm()))
        if g is not None and g.numel() <= 4096:
            rec["grad::" + k] = g.numpy()
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array(norms)
    rec["param_names"] = np.array(list(shapes.keys()))
    rec["param_shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
    rec["case"] = np.array(repr(c))
    path = os.path.join(HERE, "visual_bert_bypass.npz")
    np.savez_compressed(path, **rec)
    print("visual_bert_bypass loss", loss.item(), "scores[0,:4]", rec["scores"][0, :4], "->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
