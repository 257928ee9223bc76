"""Where the host time of the EAGER training step goes (the loop MMF's trainer drives: model(batch); loss.backward(); optimizer.step(),
mmf/trainers/core/training_loop.py:199-231): host seconds per phase without synchronisation, the synchronised step, a cProfile top list.
    python tools/eager_profile.py [--profile]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = bench.build(device, 0)
    model.train(True)
    batch = bench.synthetic_batch(32, 0, device)
    from mmf_amd.common.registry import registry
    from mmf_amd.utils.configuration import Config
    full = Config(model="visual_bert", optimizer=dict(params=dict(lr=5e-5)), model_config=dict(visual_bert=model.config))
    opt = registry.get_optimizer_class("adam_w")(model.get_optimizer_parameters(full), lr=5e-5, eps=1e-8, capturable=False)
    if "--after-graph" in sys.argv:         # what bench.py's eager leg sees: a hipGraph of the same step was captured and replayed before
        from mmf_amd.utils.graph import GraphedTrainStep
        gopt = registry.get_optimizer_class("adam_w")(model.get_optimizer_parameters(full), lr=5e-5, eps=1e-8, capturable=True)
        g = GraphedTrainStep(model, batch, warmup=2, optimizer=gopt)
        for _ in range(5):
            g()
        torch.cuda.synchronize()
        del g, gopt
    ph = {"zero_grad": 0.0, "forward": 0.0, "loss": 0.0, "backward": 0.0, "optimizer": 0.0}

    def step(acc=True):
        t = [time.perf_counter()]
        opt.zero_grad(); t.append(time.perf_counter())
        out = model(batch); t.append(time.perf_counter())
        loss = sum(v.sum() for v in out["losses"].values()); t.append(time.perf_counter())
        loss.backward(); t.append(time.perf_counter())
        opt.step(); t.append(time.perf_counter())
        if acc:
            for k, a, b in zip(ph, t, t[1:]):
                ph[k] += b - a

    for _ in range(3):
        step(False)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print("eager step: %.3f ms synchronised, %.3f ms host enqueue" % (t_all * 1e3, t_host * 1e3))
    print("host ms per phase:", {k: round(v / n * 1e3, 3) for k, v in ph.items()})
    if "--profile" in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            step(False)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
