import sys, subprocess
if len(sys.argv) == 1:
    for c in ["plain", "set_offset", "get_offset_only", "set_offset_backward"]:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True, timeout=300)
        print(c, "rc", r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1][:120], "|", " ".join(r.stderr.strip().splitlines()[-2:])[:300])
    sys.exit(0)
import torch
case = sys.argv[1]
torch.cuda.init(); torch.zeros(1, device="cuda")
gen = torch.cuda.default_generators[0]
if case.startswith("set_offset"):
    off = gen.get_offset(); gen.set_offset(off + 4)
elif case == "get_offset_only":
    off = gen.get_offset()
w = torch.randn(64, 64, device="cuda", requires_grad=True)
x = torch.randn(8, 64, device="cuda")
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        y = (x @ w).sum(); y.backward() if "backward" in case else None
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
w.grad = None
with torch.cuda.graph(g):
    y = (x @ w).sum()
    if "backward" in case: y.backward()
g.replay(); torch.cuda.synchronize()
print("ok", float(y))
