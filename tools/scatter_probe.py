"""Development aid: mmf_rows_scatter_add (word-embedding gradient shape) — the owner-wave kernel against the fp32-atomic one, isolated, hipGraph replays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmf_amd import _native as nat
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from micro_sweep import timeit

B, T, S, H, V = 32, 128, 228, 768, 30522
d = torch.randn(B * S, H, device="cuda").bfloat16()
out = torch.zeros(V, H, device="cuda")
for name, ids in (("random ids", torch.randint(1, V, (B, T), device="cuda")), ("zipf-like ids", (torch.rand(B, T, device="cuda") ** 6 * 3000).long() + 1),
                  ("half padding", torch.where(torch.rand(B, T, device="cuda") < 0.5, torch.zeros(B, T, dtype=torch.long, device="cuda"), torch.randint(1, V, (B, T), device="cuda")))):
    line = []
    for tun in (1, 0):
        nat.set_tunable(17, tun)
        us = timeit(lambda: nat.rows_scatter_add(d, H, B, T, S, ids, T, 0, 0, out, H, 0, 0))
        line.append("%s %.1f us" % ("atomics" if tun else "owner waves", us))
    nat.set_tunable(17, 0)
    print("%-14s distinct %5d of %d: %s" % (name, int(ids.unique().numel()), B * T, "   ".join(line)), flush=True)
