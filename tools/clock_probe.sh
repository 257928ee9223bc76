#!/bin/bash
# Clock / power of the chip while the graphed training step replays back to back (12 s), sampled with rocm-smi:  bash tools/clock_probe.sh [MMF_AMD_TUN value]
# (the step runs power-limited: ~1.06 kW, sclk ~2.2 GHz against 2.4 GHz nominal; profiles/r04_in_step_choices.txt)
export TMPDIR=/tmp
[ -n "$1" ] && export MMF_AMD_TUN=$1
python - <<'PY' &
import torch, time, sys, os
sys.path.insert(0, os.getcwd())
import bench
from mmf_amd.common.registry import registry
from mmf_amd.utils.configuration import Config
from mmf_amd.utils.graph import GraphedTrainStep
dev = torch.device("cuda", 0)
model = bench.build(dev, 0); model.train()
batch = bench.synthetic_batch(32, 0, dev)
full = Config(model="visual_bert", optimizer=dict(params=dict(lr=5e-5)), model_config=dict(visual_bert=model.config))
opt = registry.get_optimizer_class("adam_w")(model.get_optimizer_parameters(full), lr=5e-5, eps=1e-8, capturable=True)
g = GraphedTrainStep(model, batch, warmup=2, optimizer=opt)
print("START", flush=True)
t0=time.time(); n=0
while time.time()-t0 < 12:
    for _ in range(20): g()
    torch.cuda.synchronize(); n+=20
print("steps", n, "ms/step", (time.time()-t0)/n*1e3, flush=True)
PY
sleep 16
for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*sclk clock level: [0-9]*: //; s/.*Power (W): /W /' | tr '\n' ' '; echo; sleep 1.2; done
wait
