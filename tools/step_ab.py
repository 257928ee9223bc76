"""Same-process, interleaved A/B of the graphed training step (bench.py's headline: VisualBERT VQA2, B = 32, forward + loss + backward +
AdamW as ONE hipGraph) under several settings of the library's tunables (include/mmf_amd.h MMF_TUN_*):

    python tools/step_ab.py 14:0 14:15 [2:3,9:1 ...] [--rounds 5] [--iters 20] [--config vilbert|mmbt|m4c]

Every argument is one setting, "id:value[,id:value...]".  A tunable is read when a kernel is LAUNCHED, so each setting gets its own captured
graph (the launch arguments are frozen into it); the graphs are then replayed in turn, `iters` replays per visit, `rounds` visits each
(cdna_hip_programming.md section 5.4 rule 24: the box-to-box spread of this pool is +-6 %, only same-process numbers compare)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    args = sys.argv[1:]
    rounds, iters, settings, config = 5, 20, [], None
    i = 0
    while i < len(args):
        if args[i] == "--rounds":
            rounds = int(args[i + 1]); i += 2
        elif args[i] == "--iters":
            iters = int(args[i + 1]); i += 2
        elif args[i] == "--config":       # one of bench.py's graphed --config models (vilbert, mmbt, m4c) instead of the VisualBERT headline
            config = args[i + 1]; i += 2
        else:
            settings.append(args[i]); i += 1
    import bench
    from mmf_amd import _native as nat
    from mmf_amd.common.registry import registry
    from mmf_amd.utils.configuration import Config
    from mmf_amd.utils.graph import GraphedTrainStep
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if config is None:
        model = bench.build(dev, 0); model.train()
        batch = bench.synthetic_batch(32, 0, dev)
        full = Config(model="visual_bert", optimizer=dict(params=dict(lr=5e-5)), model_config=dict(visual_bert=model.config))
    else:
        import warnings
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
        import widened_bench as W
        from mmf_amd.common.sample import SampleList
        gen = torch.Generator().manual_seed(1234)
        torch.manual_seed(1234)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            label, B, model, sample = W.CASES[config](gen)
        model = model.to("cuda").train()
        batch = SampleList(W.to_dev(sample))
        full = Config(model=config, optimizer=dict(params=dict(lr=5e-5)), model_config={config: model.config})
    graphs = []
    for s in settings:
        items = [x.split(":") for x in s.split(",")]
        kw = {}
        kv = [(int(a), int(b)) for a, b in items]
        for k, v in kv:
            nat.set_tunable(k, v)
        opt = registry.get_optimizer_class("adam_w")(model.get_optimizer_parameters(full), lr=5e-5, eps=1e-8, capturable=True)
        g = GraphedTrainStep(model, batch, warmup=2, optimizer=opt, **kw)
        for k, v in kv:
            nat.set_tunable(k, 0)
        for _ in range(3):
            g()
        torch.cuda.synchronize()
        graphs.append((s, g, opt))
    times = {s: [] for s in settings}
    for r in range(rounds):
        for s, g, _ in graphs:
            g();
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                g()
            e1.record(); torch.cuda.synchronize()
            times[s].append(e0.elapsed_time(e1) / iters)
    for s in settings:
        t = times[s]
        print("%-24s ms/step median %.3f  min %.3f  all %s" % (s, statistics.median(t), min(t), " ".join("%.3f" % x for x in t)), flush=True)


if __name__ == "__main__":
    main()
