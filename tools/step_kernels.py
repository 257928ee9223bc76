"""Per-call-site kernel times INSIDE the training step (eager launches, HIP events around every GEMM / attention / LayerNorm call of
bench.py's VisualBERT VQA2 step), for one or several settings of the library's tunables, interleaved in one process:

    python tools/step_kernels.py 0:0 9:3 [--steps 4]

A setting is "id:value[,id:value...]" (include/mmf_amd.h MMF_TUN_*; "0:0" = defaults).  Unlike bench.py's `roofline` block (one line per
kernel family) a GEMM is keyed by its shape and epilogue, i.e. by its place in the layer: the table shows what a change does to the kernel
it touches AND to its neighbours (a producer's store policy shows up in its consumer's line).  Events add ~3 us per call (same for every
setting); the sum is therefore above the replayed graph's step time."""
import collections
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


class SiteProbe:
    def __init__(self):
        self.rec = []

    def __enter__(self):
        from mmf_amd import _native as nat
        from mmf_amd import _ops_native
        self.nat, self._ops = nat, _ops_native
        _ops_native.push_mode(1)          # the native operators' Python twins: same kernels, same launch order, entered through nat.*
        names = ("gemm", "gemm_grouped", "attention_fwd", "attention_bwd", "layernorm_fwd", "layernorm_bwd", "adamw_multi")
        self.saved = {n: getattr(nat, n) for n in names if hasattr(nat, n)}

        def wrap(name, label_of):
            orig = self.saved[name]

            def f(*a, **kw):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); r = orig(*a, **kw); e1.record()
                self.rec.append((label_of(*a, **kw), e0, e1))
                return r
            setattr(nat, name, f)

        def gemm_label(A, B, C_out, M, N, K, *a, **kw):
            epi = "+".join(x for x in ("bias" if kw.get("bias") is not None else "", {0: "", 1: "gelu", 2: "*aux", 3: "tanh", 4: "*dtanh"}[kw.get("act", 0)],
                                       "drop" if kw.get("drop") is not None and kw.get("drop") is not nat.NO_DROP else "",
                                       "resid" if kw.get("resid") is not None else "") if x)
            form = "%s%s" % ("T" if kw.get("a_kmajor") else "N", "N" if kw.get("b_kmajor") else "T")
            return "gemm %s M=%d N=%d K=%d %s | %s" % (form, M, N, K, epi or "plain", nat.gemm_last_kernel()[-24:].strip())

        wrap("gemm", gemm_label)
        wrap("gemm_grouped", lambda ps: "gemm grouped wgrad x%d | %s" % (len(ps), nat.gemm_last_kernel()[-28:].strip()))
        wrap("attention_fwd", lambda *a, **k: "attention fwd")
        wrap("attention_bwd", lambda *a, **k: "attention bwd")
        wrap("layernorm_fwd", lambda *a, **k: "layernorm fwd")
        wrap("layernorm_bwd", lambda *a, **k: "layernorm bwd")
        return self

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(self.nat, n, f)
        self._ops.pop_mode(1)

    def table(self):
        torch.cuda.synchronize()
        by = collections.OrderedDict()
        for label, e0, e1 in self.rec:
            by.setdefault(label, []).append(e0.elapsed_time(e1) * 1e3)
        return by


def main():
    args = sys.argv[1:]
    steps, settings = 4, []
    i = 0
    while i < len(args):
        if args[i] == "--steps":
            steps = int(args[i + 1]); i += 2
        else:
            settings.append(args[i]); i += 1
    settings = settings or ["0:0"]
    import bench
    from mmf_amd import _native as nat
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model = bench.build(dev, 0); model.train()
    batch = bench.synthetic_batch(32, 0, dev)

    def step():
        model.zero_grad(set_to_none=True)
        out = model(batch)
        sum(v.sum() for v in out["losses"].values()).backward()

    side = torch.cuda.Stream(device=dev)
    results = {s: collections.OrderedDict() for s in settings}
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
        for r in range(steps):
            for s in settings:
                kv = [(int(a), int(b)) for a, b in (x.split(":") for x in s.split(","))]
                for k, v in kv:
                    nat.set_tunable(k, v)
                with SiteProbe() as p:
                    step()
                    t = p.table()
                for k, v in kv:
                    nat.set_tunable(k, 0)
                for label, us in t.items():
                    results[s].setdefault(label, []).append((len(us), sum(us)))
    labels = []
    for s in settings:
        for l in results[s]:
            if l not in labels:
                labels.append(l)
    print("%-78s %s" % ("call site (launches per step)", "  ".join("%16s" % s for s in settings)))
    tot = {s: 0.0 for s in settings}
    for l in labels:
        row, n = [], 0
        for s in settings:
            v = results[s].get(l)
            if v is None:
                row.append("%16s" % "-")
                continue
            n = v[0][0]
            per = statistics.median(x[1] / x[0] for x in v)
            tot[s] += per * n
            row.append("%9.1f us ea." % per)
        print("%-72s x%-4d %s" % (l[:72], n, "  ".join(row)))
    print("%-78s %s" % ("sum over the probed calls (ms per step, events included)", "  ".join("%13.3f ms" % (tot[s] / 1e3) for s in settings)))


if __name__ == "__main__":
    main()
