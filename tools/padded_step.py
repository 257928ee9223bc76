"""The graphed VisualBERT VQA2 training step (bench.py's model, B = 32, AdamW inside the graph) on TRIMMED batches: text lengths cut to the columns the batch
uses (mmf_amd/common/prefetch.py::trim_text_padding), one captured step per length (mmf_amd/utils/graph.py::BucketedTrainStep).

    python tools/padded_step.py [24 32 64 128 ...] [--iters 20] [--trace-only T] [--tun id:value,...]

Prints ms per step for each text length (positions per sample = T + 100).  `--trace-only T` replays only that length (for rocprofv3 --kernel-trace +
tools/trace_agg.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    args = sys.argv[1:]
    iters, lens, only = 20, [], None
    i = 0
    while i < len(args):
        if args[i] == "--iters":
            iters = int(args[i + 1]); i += 2
        elif args[i] == "--trace-only":
            only = int(args[i + 1]); i += 2
        elif args[i] == "--tun":          # "id:value[,id:value]": library tunables (include/mmf_amd.h MMF_TUN_*) set before the captures
            from mmf_amd import _native as nat
            for kv in args[i + 1].split(","):
                k, v = kv.split(":"); nat.set_tunable(int(k), int(v))
            i += 2
        else:
            lens.append(int(args[i])); i += 1
    lens = lens or [24, 32, 48, 64, 96, 128]
    if only is not None:
        lens = [only]
    import bench
    from mmf_amd.common.registry import registry
    from mmf_amd.utils.configuration import Config
    from mmf_amd.utils.graph import BucketedTrainStep
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model = bench.build(dev, 0); model.train()
    full = Config(model="visual_bert", optimizer=dict(params=dict(lr=5e-5)), model_config=dict(visual_bert=model.config))
    opt = registry.get_optimizer_class("adam_w")(model.get_optimizer_parameters(full), lr=5e-5, eps=1e-8, capturable=True)
    step = BucketedTrainStep(model, optimizer=opt, warmup=2, trim=1)
    batches = {}
    for T in lens:
        b = bench.synthetic_batch(32, 0, None)
        keep = (torch.arange(128)[None, :] < T).long().expand(32, 128).contiguous()
        b["input_mask"] = keep
        b["input_ids"] = b["input_ids"] * keep
        batches[T] = b.to(dev)
        step(batches[T])
    torch.cuda.synchronize()
    for rnd in range(3 if only is None else 1):
        for T in lens:
            from mmf_amd.common.prefetch import trim_text_padding
            tb = trim_text_padding(batches[T], 1)
            for _ in range(3):
                step(tb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                step(tb)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / iters * 1e3
            print("T = %3d  positions %3d  rows %5d   %.3f ms per step   %.0f samples/s" % (T, T + 100, 32 * (T + 100), dt, 32e3 / dt), flush=True)


if __name__ == "__main__":
    main()
