"""The N > 1 path on the GPU box (one MI355X): two ranks share the device over gloo (`MMF_AMD_DIST_BACKEND=gloo`; RCCL refuses
duplicate devices) and run the HIP-backed VisualBERT on halves of a batch; the GradientReducer must leave both ranks with the
gradients of the whole batch (the mean of the two ranks' gradients), keep the never-used BertPooler without a gradient, freeze
the used-parameter set after two steps and keep working after that.  Reference: DistributedDataParallel at
mmf/trainers/core/device.py:104-110 with `find_unused_parameters` (tools/sweeps/sweep_visual_bert.py:41)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _batch4(sample):
    """Four samples out of the fixture's three: rows 0, 1, 2, 0."""
    idx = torch.tensor([0, 1, 2, 0])
    out = {}
    for k, v in sample.items():
        if isinstance(v, torch.Tensor):
            out[k] = v[idx].clone()
        elif isinstance(v, dict):
            out[k] = {kk: vv[idx].clone() for kk, vv in v.items()}
        else:
            out[k] = v
    return out


def _half(batch, rank):
    sl = slice(2 * rank, 2 * rank + 2)
    out = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            out[k] = v[sl].clone()
        elif isinstance(v, dict):
            out[k] = {kk: vv[sl].clone() for kk, vv in v.items()}
        else:
            out[k] = v
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MMF_AMD_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from mmf_amd.common.sample import SampleList
    from mmf_amd.trainers.core.device import parallelize_model
    from mmf_amd.utils import distributed as D
    from tests.golden_utils import load_case
    from tests.model_utils import build_visual_bert, sample_to
    D.distributed_init_from_env()
    z, case, cfg, sd, sample = load_case("small64")
    model = build_visual_bert(cfg, sd)
    model.eval()
    reducer = parallelize_model(model, bucket_bytes=1 << 18, comm_dtype=torch.bfloat16)       # several buckets even for the small model; bf16 wire (opt-in)
    full = _batch4(sample)
    mine = SampleList(sample_to(_half(full, rank), "cuda"))
    frozen = []
    for step in range(3):
        model.zero_grad(set_to_none=True)
        out = model(mine)
        sum(v.sum() for v in out["losses"].values()).backward()
        reducer.finish()
        frozen.append(reducer._frozen)
    got = {n: (p.grad.detach().float().cpu().numpy() if p.grad is not None else None) for n, p in model.named_parameters()}      # (numpy: pickled by value)
    ref = None
    if rank == 0:       # the whole batch on one rank; no_sync(): this backward must not launch collectives nobody else joins
        model.zero_grad(set_to_none=True)
        with reducer.no_sync():
            out = model(SampleList(sample_to(full, "cuda")))
            sum(v.sum() for v in out["losses"].values()).backward()
        ref = {n: (p.grad.detach().float().cpu().numpy() if p.grad is not None else None) for n, p in model.named_parameters()}
    wire = sorted({str(b["dtype"]) for b in reducer.buckets})
    q.put((rank, got, ref, frozen, wire, len(reducer.buckets)))
    D.synchronize()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reduce_to_the_whole_batch_gradient():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            rank, got, ref, frozen, wire, nb = q.get(timeout=150)
            res[rank] = (got, ref, frozen, wire, nb)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:         # never leave a rank stuck in a collective behind
            if p.is_alive():
                p.kill()
    ref = res[0][1]
    assert res[0][3] == ["torch.bfloat16", "torch.float32"] and res[0][4] >= 3          # bf16 buckets + fp32 embedding buckets
    assert res[0][2] == [False, True, True] and res[1][2] == [False, True, True]         # used set frozen after it repeated once
    checked = 0
    for name, r in ref.items():
        g0, g1 = res[0][0][name], res[1][0][name]
        if r is None:
            assert g0 is None and g1 is None, name          # the pooler under `vqa`: unused everywhere, stays None
            continue
        assert g0 is not None and g1 is not None, name
        g0, g1, r = torch.from_numpy(g0), torch.from_numpy(g1), torch.from_numpy(r)
        assert torch.equal(g0, g1), name                    # both ranks hold the same averaged gradient
        scale = float(r.abs().max()) + 1e-12
        tol = 3e-2 if "embeddings" not in name else 1e-2    # bf16 on the wire (fp32 for the embedding tables) + bf16 kernels at B = 2 vs 4
        assert float((g0 - r).abs().max()) <= tol * scale, (name, float((g0 - r).abs().max()), scale)
        checked += 1
    assert checked > 30
