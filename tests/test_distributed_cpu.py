"""world_size-2 gloo tests (CPU) of the data-parallel path: the GradientReducer must leave every rank
with the mean gradient, tolerate parameters that never receive one (the BertPooler case) and keep
working across steps; reduce_dict / broadcast_scalar follow the reference helpers."""
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
    reducer = GradientReducer(model, bucket_bytes=256)   # tiny buckets -> several collectives
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


def test_single_process_is_a_noop():
    from mmf_amd.trainers.core.device import parallelize_model
    m = torch.nn.Linear(2, 2)
    red = parallelize_model(m)
    m(torch.ones(1, 2)).sum().backward()
    g = m.weight.grad.clone()
    red.finish()
    assert torch.equal(m.weight.grad, g)
