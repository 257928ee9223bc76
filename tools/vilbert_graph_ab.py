"""ViLBERT (BASELINE configs[2] share of one GPU, B = 32) as ONE hipGraph per training step, with the visual stream on a side HIP stream
(parallel graph branches) or on the caller's stream: python tools/vilbert_graph_ab.py   (env MMF_AMD_VILBERT_STREAMS=0 / 1)."""
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

import widened_bench as W
from mmf_amd.common.registry import registry
from mmf_amd.common.sample import SampleList
from mmf_amd.utils.configuration import Config
from mmf_amd.utils.graph import GraphedTrainStep

name = sys.argv[1] if len(sys.argv) > 1 else "vilbert"
g = torch.Generator().manual_seed(1234)
torch.manual_seed(1234)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    label, B, model, sample = W.CASES[name](g)
model = model.to("cuda").train()
batch = SampleList(W.to_dev(sample))
full = Config(model=name, optimizer=dict(params=dict(lr=5e-5)), model_config={name: model.config})
opt = registry.get_optimizer_class("adam_w")(model.get_optimizer_parameters(full), lr=5e-5, eps=1e-8, capturable=True)
step = GraphedTrainStep(model, batch, warmup=2, optimizer=opt)
for _ in range(3):
    step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n * 1e3
print("%s streams=%s graph step %.3f ms  (%.1f samples/s)  loss %.4f" % (name, os.environ.get("MMF_AMD_VILBERT_STREAMS", "1"), dt, B * 1e3 / dt, float(loss)))
