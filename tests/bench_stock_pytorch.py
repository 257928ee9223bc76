"""Comparator, not a test: the reference algorithm (the pinned oracle = the reference's own op sequence) executed by STOCK
PyTorch-ROCm on the same MI355X (rocBLAS / hipBLASLt GEMMs, ATen softmax / LayerNorm / dropout kernels), fp32 and
bf16 autocast, forward + logit_bce + backward at the benchmark shape (SURVEY.md §8d "the honest comparator").

    python tests/bench_stock_pytorch.py        # on the GPU box; writes gpurun_out/stock_pytorch.json

Lives under tests/ because only tests may import oracle/ (tests/test_oracle_isolation.py)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import visual_bert_oracle as O  # noqa: E402


def to_dev(x, dev):
    if isinstance(x, torch.Tensor):
        return x.to(dev)
    if isinstance(x, dict):
        return {k: to_dev(v, dev) for k, v in x.items()}
    return x


def run(autocast, B=32, steps=10, warmup=3):
    dev = torch.device("cuda")
    cfg = dict(O.DEFAULT_CONFIG)
    sd = {k: v.to(dev).requires_grad_(True) for k, v in O.init_state_dict(cfg, seed=1234).items()}
    sample = to_dev(O.synthetic_batch(cfg, B, seed=1234), dev)

    def one():
        for v in sd.values():
            v.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            out = O.train_step_loss(sd, cfg, sample, train=True)
        list(out["losses"].values())[0].backward()

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"ms_per_step": round(dt * 1e3, 3), "samples_per_s": round(B / dt, 1)}


if __name__ == "__main__":
    res = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "workload": "VisualBERT VQA2 fwd+logit_bce+bwd, B=32, train mode",
           "fp32": run(False), "bf16_autocast": run(True)}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/stock_pytorch.json", "w"), indent=1)
    print(json.dumps(res))
