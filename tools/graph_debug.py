import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = ["small_eval_fwdonly", "small_eval_fwdonly_sidestream", "small_eval_fwdonly_nograd"]
if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True, timeout=300)
        print(c, "rc", r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1][:200], (r.stderr.strip().splitlines() or ["-"])[-1][:200] if r.returncode else "")
    sys.exit(0)
case = sys.argv[1]
import torch
from tests.golden_utils import load_case
from tests.model_utils import build_visual_bert, sample_to
from mmf_amd.common.sample import SampleList
from mmf_amd.utils.graph import GraphedTrainStep
from oracle import visual_bert_oracle as O
if case.startswith("small"):
    z, cs, cfg, sd, sample = load_case("small64")
else:
    cfg = dict(O.DEFAULT_CONFIG); cfg["num_hidden_layers"] = 2
    sd = O.init_state_dict(cfg, seed=7); sample = O.synthetic_batch(cfg, 2, seed=99)
model = build_visual_bert(cfg, sd)
batch = SampleList(sample_to(sample, "cuda"))
model.train("train" in case)
import gc
import contextlib
ctx = torch.cuda.stream(torch.cuda.Stream()) if "sidestream" in case else contextlib.nullcontext()
ng = torch.no_grad() if "nograd" in case else contextlib.nullcontext()
if "noprior" not in case:
  with ctx, ng:
    out = model(batch); loss = sum(v.sum() for v in out["losses"].values())
    if "fwdonly" not in case:
        loss.backward()
    if "cleanup" in case:
        del out, loss; model.zero_grad(set_to_none=True); gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    if "sync" in case:
        torch.cuda.synchronize()
print("prior ok", flush=True)
g = GraphedTrainStep(model, batch, warmup=0 if "nowarm" in case else 2)
print("captured", flush=True)
print("replay", float(g()), float(g()), flush=True)
