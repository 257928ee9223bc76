"""Where the graphed training step spends time that is NOT inside a kernel.

    rocprofv3 --kernel-trace --output-format csv -d OUT -o g -- python tools/graph_gaps.py run     # on the GPU box
    python tools/graph_gaps.py report OUT/g_kernel_trace.csv                                       # anywhere

`run` replays bench.py's step (VisualBERT VQA2, B = 32, forward + loss + backward + AdamW as ONE hipGraph) 12 times.  `report` cuts
the kernel trace at the graph's first node (`step_advance_kernel`; `seed_advance_kernel` before round 5), and for the last replays prints wall time per replay, the sum
of kernel durations, their difference (idle gaps between dependent graph nodes) and the kernels grouped by name with launch counts:
the gap total divided by the node count is the cost of ONE more kernel in the graph, i.e. what fusing a tiny kernel away buys."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from bench import build, synthetic_batch
    from mmf_amd.common.registry import registry
    from mmf_amd.utils.configuration import Config
    from mmf_amd.utils.graph import GraphedTrainStep
    dev = torch.device("cuda", 0)
    for kv in os.environ.get("GRAPH_GAPS_TUN", "").split(","):      # "id:value,..." (include/mmf_amd.h MMF_TUN_*): A/B traces of one tunable
        if kv:
            from mmf_amd import _native as nat
            nat.set_tunable(int(kv.split(":")[0]), int(kv.split(":")[1]))
    model = build(dev, 0)
    model.train()
    batch = synthetic_batch(32, 0, dev)
    full = Config(model="visual_bert", optimizer=dict(params=dict(lr=5e-5)), model_config=dict(visual_bert=model.config))
    opt = registry.get_optimizer_class("adam_w")(model.get_optimizer_parameters(full), lr=5e-5, eps=1e-8, capturable=True)
    g = GraphedTrainStep(model, batch, warmup=2, optimizer=opt)
    for _ in range(12):
        g()
    torch.cuda.synchronize()


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    for cut in ("<", "("):
        if cut in name and not name.startswith("at::"):
            name = name.split(cut)[0]
    if name.startswith("at::native::"):
        name = "torch:" + name.split("<")[0].split("::")[-1] + ("<" + name.split("<")[1].split(",")[0].split("::")[-1][:40] if "<" in name else "")
    return name[:70]


def report(path, out_json=None):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if ("seed_advance_kernel" in r[2] or "step_advance_kernel" in r[2])]
    # the last 13 cuts delimit the 12 replays (earlier ones are the eager warm-up passes and the capture)
    cuts = cuts[-12:]
    steps = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        ks = rows[a:b]
        wall = ks[-1][1] - ks[0][0]
        busy = sum(e - s for s, e, _ in ks)
        # overlap-aware busy time (union of intervals)
        union, cur_s, cur_e = 0, ks[0][0], ks[0][1]
        for s, e, _ in ks[1:]:
            if s > cur_e:
                union += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        union += cur_e - cur_s
        steps.append((wall, busy, union, len(ks), rows[b][0] - ks[0][0]))
    steps = steps[2:]
    n = len(steps)
    wall = sum(s[0] for s in steps) / n / 1e3
    busy = sum(s[1] for s in steps) / n / 1e3
    union = sum(s[2] for s in steps) / n / 1e3
    period = sum(s[4] for s in steps) / n / 1e3
    nodes = steps[-1][3]
    by = collections.OrderedDict()
    a, b = cuts[-2], cuts[-1]
    for s, e, name in rows[a:b]:
        d = by.setdefault(short(name), [0, 0])
        d[0] += 1
        d[1] += e - s
    res = {"replays_averaged": n, "kernels_per_replay": nodes, "replay_period_us": round(period, 1), "first_to_last_kernel_us": round(wall, 1),
           "sum_of_kernel_durations_us": round(busy, 1), "busy_union_us": round(union, 1), "idle_gaps_us": round(wall - union, 1),
           "idle_per_kernel_us": round((wall - union) / nodes, 2),
           "kernels": [{"name": k, "launches": v[0], "us": round(v[1] / 1e3, 1)} for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])]}
    small = [k for k in res["kernels"] if k["us"] / k["launches"] < 8.0]
    res["kernels_under_8us"] = {"launches": sum(k["launches"] for k in small), "us": round(sum(k["us"] for k in small), 1)}
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
    for k in res["kernels"]:
        print("%5d  %9.1f us  %7.2f us each  %s" % (k["launches"], k["us"], k["us"] / k["launches"], k["name"]))
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)
    return res


def timeline(path):
    """Start offset / duration / name of every kernel of the LAST replay, in start order (overlapping branches show as overlapping rows)."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if ("seed_advance_kernel" in r[2] or "step_advance_kernel" in r[2])]
    a, b = cuts[-2], cuts[-1]
    t0 = rows[a][0]
    prev_end = t0
    for s, e, name in rows[a:b]:
        print("%9.1f  %8.1f us  %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, "|| " if s < prev_end - 500 else "   ", short(name)))
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2])
    else:
        report(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
