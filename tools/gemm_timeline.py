"""Where a GEMM launch spends its time INSIDE the training step: runs the VisualBERT VQA2 step eagerly with the GEMM timeline
probe on (mmf_gemm_set_probe: one record of s_memtime stamps per workgroup) and prints, per launch shape, the workgroup-level
breakdown — prologue (entry -> first stage landed), K loop, accumulator staging, global epilogue + store drain — and the launch
envelope (first entry -> last exit).  python tools/gemm_timeline.py [--batch 32]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from mmf_amd import _native as nat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--mhz", type=float, default=100.0, help="s_memtime tick rate (MHz) used to print microseconds")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = bench.build(dev, 0)
    model.train(True)
    batch = bench.synthetic_batch(args.batch, 0, dev)

    def step():
        model.zero_grad(set_to_none=True)
        out = model(batch)
        sum(v.sum() for v in out["losses"].values()).backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    # calibrate the tick: a probed launch bracketed by HIP events
    cap = 400000
    buf = torch.zeros(8 * (1 + cap), dtype=torch.int64, device=dev)

    nat.gemm_set_probe(buf)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    step()
    e1.record()
    torch.cuda.synchronize()
    nat.gemm_set_probe(None)
    LAY = {0: "NT", 1: "TT", 2: "NN", 3: "TN"}
    shapes = {l: (("grouped " + LAY[lay & 3]) if lay & 4 else LAY[lay], M_, N_, K_) for l, lay, M_, N_, K_ in nat.gemm_probe_log()}
    n = min(int(buf[0].item()), cap)
    rec = buf[8:8 * (1 + n)].view(n, 8).cpu()
    launch = (rec[:, 0] >> 32).tolist()
    by = collections.defaultdict(list)
    for i, l in enumerate(launch):
        by[l].append(i)
    span_ticks = int(rec[:, 6].max() - rec[:, 2].min())
    step_ms = e0.elapsed_time(e1)
    tick_us = 1.0 / args.mhz
    print("records %d, launches %d, eager probed step %.2f ms by HIP events, first GEMM entry -> last GEMM exit %.2f ms by the probe clock"
          % (n, len(by), step_ms, span_ticks * tick_us / 1e3))
    agg = collections.OrderedDict()
    for l in sorted(by):
        idx = by[l]
        r = rec[idx].double()
        key = shapes.get(l, ("?",)) + (len(idx),)
        t0 = r[:, 2].min()
        env = (r[:, 6].max() - t0) * tick_us
        pro = ((r[:, 3] - r[:, 2]) * tick_us).mean(); kl = ((r[:, 4] - r[:, 3]) * tick_us).mean()
        stg = ((r[:, 5] - r[:, 4]) * tick_us).mean(); epi = ((r[:, 6] - r[:, 5]) * tick_us).mean()
        start_spread = ((r[:, 2] - t0) * tick_us)
        a = agg.setdefault(key, [])
        a.append((env.item(), pro.item(), kl.item(), stg.item(), epi.item(), len(idx), start_spread.max().item()))
    print("%-28s %5s %6s | %8s | %7s %7s %7s %7s | %s" % ("launch (kind, M, N, K)", "count", "wgs", "envelope", "prolog", "k-loop", "stage", "epi+st", "last wg start"))
    for key, v in agg.items():
        t = torch.tensor(v, dtype=torch.float64)
        m = t.mean(0)
        print("%-28s %5d %6d | %8.1f | %7.2f %7.2f %7.2f %7.2f | %6.1f   (us)" % (str(key), len(v), int(m[5]), m[0], m[1], m[2], m[3], m[4], m[6]))


if __name__ == "__main__":
    main()
