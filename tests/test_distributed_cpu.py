"""world_size-2 gloo tests (CPU) of the data-parallel path: the GradientReducer must leave every rank
with the mean gradient, tolerate parameters that never receive one (the BertPooler case) and keep
working across steps; reduce_dict / broadcast_scalar follow the reference helpers."""
import contextlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mmf_amd.trainers.core.device import GradientReducer
    from mmf_amd.utils import distributed as D
    r, w = D.distributed_init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
    unused = torch.nn.Linear(3, 3)            # never used in forward: gets no gradient
    model.add_module("unused", unused)
    reducer = GradientReducer(model, bucket_bytes=256, comm_dtype=torch.float32)   # tiny buckets -> several collectives
    out = {}
    for step in range(3):
        model.zero_grad(set_to_none=True)
        g = torch.Generator().manual_seed(100 * step + rank)
        x = torch.randn(5, 8, generator=g)
        loss = model[2](model[1](model[0](x))).pow(2).sum()
        loss.backward()
        local = [p.grad.tolist() if p.grad is not None else None for p in model.parameters()]
        reducer.finish()
        out[step] = (local, [p.grad.tolist() if p.grad is not None else None for p in model.parameters()])
    red = D.reduce_dict({"a": torch.tensor(float(rank + 1)), "b": torch.tensor(2.0)})
    bs = D.broadcast_scalar(41 + rank, src=0)
    q.put((rank, out, {k: float(v) for k, v in red.items()}, bs))
    D.synchronize()
    dist.destroy_process_group()


def test_gradient_reducer_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, out, red, bs = q.get(timeout=120)
        res[rank] = (out, red, bs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step in range(3):
        l0, g0 = res[0][0][step]
        l1, g1 = res[1][0][step]
        for a, b, ra, rb in zip(l0, l1, g0, g1):
            if a is None:
                assert b is None and ra is None and rb is None
                continue
            mean = (torch.tensor(a) + torch.tensor(b)) / 2
            assert torch.allclose(torch.tensor(ra), mean, atol=1e-6) and torch.allclose(torch.tensor(rb), mean, atol=1e-6)
    assert res[0][1] == {"a": 1.5, "b": 2.0}          # rank 0 holds the mean (reference reduce_dict semantics)
    assert res[0][2] == 41 and res[1][2] == 41


def _worker2(rank, world, port, q):
    """Accumulation under no_sync(), a parameter used on ONE rank only, bf16 buckets with an fp32 embedding bucket, freezing."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mmf_amd.trainers.core.device import GradientReducer
    from mmf_amd.utils import distributed as D
    D.distributed_init_from_env(backend="gloo")
    torch.manual_seed(0)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb = torch.nn.Embedding(11, 8)
            self.a = torch.nn.Linear(8, 16)
            self.side = torch.nn.Linear(8, 16)     # used by rank 1 only, from step 2 on
            self.b = torch.nn.Linear(16, 4)
            self.never = torch.nn.Linear(3, 3)

        def forward(self, ids, use_side):
            h = self.emb(ids).mean(1)
            z = self.a(h)
            if use_side:
                z = z + self.side(h)
            return self.b(torch.tanh(z))

    net = Net()
    red = GradientReducer(net, bucket_bytes=512, static_graph=False, comm_dtype=torch.bfloat16)
    dts = sorted({str(b["dtype"]) for b in red.buckets})
    emb_bucket = red.buckets[red._bucket_of[id(net.emb.weight)]]
    out = {"dtypes": dts, "emb_fp32_alone": emb_bucket["dtype"] == torch.float32 and all(p is net.emb.weight for p in emb_bucket["params"])}
    res = {}
    for step in range(4):
        net.zero_grad(set_to_none=True)
        use_side = (rank == 1 and step >= 2)
        micro = []
        for m in range(2):       # two micro-batches: the first accumulates locally, the second synchronises
            g = torch.Generator().manual_seed(1000 * step + 10 * m + rank)
            ids = torch.randint(0, 11, (6, 3), generator=g)
            ctx = red.no_sync() if m == 0 else contextlib.nullcontext()
            with ctx:
                net(ids, use_side).pow(2).sum().backward()
                red.finish()
            micro.append(None)
        res[step] = {n: (p.grad.clone() if p.grad is not None else None) for n, p in net.named_parameters()}
    # reference: the same computation without any reducer, gathered through the queue
    ref = {}
    torch.manual_seed(0)
    net2 = Net()
    for step in range(4):
        net2.zero_grad(set_to_none=True)
        use_side = (rank == 1 and step >= 2)
        for m in range(2):
            g = torch.Generator().manual_seed(1000 * step + 10 * m + rank)
            ids = torch.randint(0, 11, (6, 3), generator=g)
            net2(ids, use_side).pow(2).sum().backward()
        ref[step] = {n: (p.grad.clone() if p.grad is not None else None) for n, p in net2.named_parameters()}
    q.put((rank, out, {k: {n: (None if v is None else v.tolist()) for n, v in d.items()} for k, d in res.items()},
           {k: {n: (None if v is None else v.tolist()) for n, v in d.items()} for k, d in ref.items()}))
    D.synchronize()
    dist.destroy_process_group()


def test_gradient_reducer_accumulation_partial_use_and_wire_dtypes():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker2, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, out, res, ref = q.get(timeout=180)
        got[rank] = (out, res, ref)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0]["dtypes"] == ["torch.bfloat16", "torch.float32"] and got[0][0]["emb_fp32_alone"]
    for step in range(4):
        for name in got[0][1][step]:
            r0, r1 = got[0][1][step][name], got[1][1][step][name]
            l0, l1 = got[0][2][step][name], got[1][2][step][name]
            if l0 is None and l1 is None:
                assert r0 is None and r1 is None, (step, name)       # unused everywhere: stays None (the optimizer skips it)
                continue
            z = torch.zeros_like(torch.tensor(l0 if l0 is not None else l1))
            mean = ((torch.tensor(l0) if l0 is not None else z) + (torch.tensor(l1) if l1 is not None else z)) / 2
            tol = 1e-6 if name.startswith("emb") else 2e-2 * float(mean.abs().max()) + 1e-6      # bf16 on the wire elsewhere
            assert r0 is not None and r1 is not None, (step, name)
            assert torch.allclose(torch.tensor(r0), mean, atol=tol) and torch.allclose(torch.tensor(r1), mean, atol=tol), (step, name)


def _worker3(rank, world, port, q):
    """(i) a parameter whose WHOLE bucket was skipped (nobody used any of its parameters in the previous step) and that
    then fires on ONE rank only must come out as the mean on both ranks (not local + mean, not a TypeError on the rank where it did
    not fire); (ii) static_graph=True must not freeze — and must not raise later — while some parameter is used on one rank only."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mmf_amd.trainers.core.device import GradientReducer
    from mmf_amd.utils import distributed as D
    D.distributed_init_from_env(backend="gloo")
    torch.manual_seed(0)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(8, 16)
            self.side = torch.nn.Linear(8, 16)      # its weight and bias fill buckets of their own (bucket_bytes is tiny)
            self.b = torch.nn.Linear(16, 4)

        def forward(self, x, use_side):
            z = self.a(x)
            if use_side:
                z = z + self.side(x)
            return self.b(torch.tanh(z))

    res, frozen, err = {}, {}, None
    for static in (False, True):
        torch.manual_seed(0)
        net = Net()
        red = GradientReducer(net, bucket_bytes=64, static_graph=static)
        try:
            for step in range(6):
                net.zero_grad(set_to_none=True)
                # static_graph=False: steps 0-1 side unused everywhere (its buckets are skipped), from step 2 on used on rank 1 only;
                # static_graph=True: used on rank 1 only from the first step on (a use pattern that CHANGES after a legitimate
                # freeze raises by contract, like DDP's static_graph)
                use_side = rank == 1 and (static or step >= 2)
                g = torch.Generator().manual_seed(100 * step + rank)
                x = torch.randn(5, 8, generator=g)
                net(x, use_side).pow(2).sum().backward()
                local = {n: (p.grad.clone() if p.grad is not None else None) for n, p in net.named_parameters()}
                red.finish()
                res[(static, step)] = (local, {n: (p.grad.clone() if p.grad is not None else None) for n, p in net.named_parameters()})
            frozen[static] = red._frozen
        except Exception as e:      # noqa: BLE001 — reported to the parent, which fails the test
            err = repr(e)
        red.remove()
    ser = {"%d_%d" % (int(k[0]), k[1]): tuple({n: (None if t is None else t.tolist()) for n, t in d.items()} for d in v) for k, v in res.items()}
    q.put((rank, ser, frozen, err))
    D.synchronize()
    dist.destroy_process_group()


def test_gradient_reducer_skipped_bucket_straggler_and_rankwise_use_never_freeze():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker3, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, res, frozen, err = q.get(timeout=180)
        got[rank] = (res, frozen, err)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][2] is None and got[1][2] is None, (got[0][2], got[1][2])
    assert got[0][1] == {False: False, True: False} and got[1][1] == {False: False, True: False}     # rank-dependent use: never frozen
    for key in got[0][0]:
        (l0, r0), (l1, r1) = got[0][0][key], got[1][0][key]
        for name in l0:
            if l0[name] is None and l1[name] is None:
                assert r0[name] is None and r1[name] is None, (key, name)
                continue
            z = torch.zeros_like(torch.tensor(l0[name] if l0[name] is not None else l1[name]))
            mean = ((torch.tensor(l0[name]) if l0[name] is not None else z) + (torch.tensor(l1[name]) if l1[name] is not None else z)) / 2
            assert r0[name] is not None and r1[name] is not None, (key, name)
            assert torch.allclose(torch.tensor(r0[name]), mean, atol=1e-6), (key, name)
            assert torch.allclose(torch.tensor(r1[name]), mean, atol=1e-6), (key, name)


def test_single_process_is_a_noop():
    from mmf_amd.trainers.core.device import parallelize_model
    m = torch.nn.Linear(2, 2)
    red = parallelize_model(m)
    m(torch.ones(1, 2)).sum().backward()
    g = m.weight.grad.clone()
    red.finish()
    assert torch.equal(m.weight.grad, g)
