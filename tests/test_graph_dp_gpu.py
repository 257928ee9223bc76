"""The data-parallel training step as a chain of hipGraphs (mmf_amd/utils/graph.py GraphedDataParallelStep): forward, backward cut
into stages at encoder layers, optimizer — with the gradient all-reduces launched between the stage graphs.

  * one rank: the chained graphs must train exactly like the eager step (same kernels, same order inside every segment);
  * two ranks sharing the GPU over gloo: both ranks end every step with the same parameters, and the update is the update
    of the whole batch (mean gradient) up to bf16 on the wire.
Reference for the semantics: DistributedDataParallel at mmf/trainers/core/device.py:104-110."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _layers(model):
    return [m for m in model.modules() if type(m).__name__ == "BertLayerJit"]


def _optimizer(model, capturable):
    from mmf_amd.modules.optimizers import AdamW
    return AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, eps=1e-8, weight_decay=0.01, capturable=capturable)


def test_chained_graphs_train_like_the_eager_step():
    from mmf_amd.common.sample import SampleList
    from mmf_amd.utils.graph import GraphedDataParallelStep
    from tests.golden_utils import load_case
    from tests.model_utils import build_visual_bert, sample_to
    z, case, cfg, sd, sample = load_case("small64")
    batch = SampleList(sample_to(sample, "cuda"))
    eager = build_visual_bert(cfg, sd); eager.eval()           # eval: no dropout, the two runs see the same arithmetic
    chained = build_visual_bert(cfg, sd); chained.eval()
    opt_e, opt_c = _optimizer(eager, False), _optimizer(chained, True)
    step = GraphedDataParallelStep(chained, batch, _layers(chained), opt_c, warmup=2)
    assert len(step.g_bwd) == 3 and [len(s) > 0 for s in step.stage_params] == [True, True, True]
    names = {id(p): n for n, p in chained.named_parameters()}
    assert all("layer.1" in names[id(p)] for p in step.stage_params[1])           # the middle stage is exactly encoder layer 1
    assert any("word_embeddings" in names[id(p)] for p in step.stage_params[2])
    unused = [n for n, p in chained.named_parameters() if not any(id(p) in {id(q) for q in s} for s in step.stage_params)]
    assert unused and all("pooler" in n for n in unused), unused                  # `vqa` pooling never touches BertPooler
    losses_e, losses_c = [], []
    for _ in range(3):
        eager.zero_grad(set_to_none=True)
        out = eager(batch)
        loss = sum(v.sum() for v in out["losses"].values())
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss))
        losses_c.append(float(step()))
    assert losses_c == pytest.approx(losses_e, rel=1e-5), (losses_c, losses_e)
    assert losses_e[2] < losses_e[0]
    pe, pc = dict(eager.named_parameters()), dict(chained.named_parameters())
    for n in pe:
        d = float((pe[n].detach() - pc[n].detach()).abs().max())
        assert d <= 1e-6 + 1e-5 * float(pe[n].detach().abs().max()), (n, d)
    assert int(round(float(opt_c._dev_state[0]))) == 3                            # the warm-up ran no optimizer step


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MMF_AMD_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from mmf_amd.common.sample import SampleList
    from mmf_amd.utils import distributed as D
    from mmf_amd.utils.graph import GraphedDataParallelStep
    from tests.golden_utils import load_case
    from tests.model_utils import build_visual_bert, sample_to
    from tests.test_distributed_gpu import _batch4, _half
    D.distributed_init_from_env()
    z, case, cfg, sd, sample = load_case("small64")
    full = _batch4(sample)
    model = build_visual_bert(cfg, sd); model.eval()
    opt = _optimizer(model, True)
    step = GraphedDataParallelStep(model, SampleList(sample_to(_half(full, rank), "cuda")), _layers(model), opt, warmup=1)
    wires = sorted({str(b[k].dtype) for b in step.buckets for k in ("wire16", "wire32") if b[k] is not None})
    names = {id(p): n for n, p in model.named_parameters()}
    wires.append("rows:" + ",".join(sorted(names[id(sp.p)] for sp in step.sparse)))       # what travels as touched rows (round 5)
    step()
    torch.cuda.synchronize()
    # after the step the wire buffers hold the SUM over ranks (the 1 / world lives in optimizer.grad_scale); the optimizer read them there
    red = step.gradients()
    grads = {n: (red[p].detach().float().cpu().numpy() if p in red else None) for n, p in model.named_parameters()}
    step()
    torch.cuda.synchronize()
    got = {n: p.detach().float().cpu().numpy() for n, p in model.named_parameters()}
    got["__grads__"] = grads
    ref = None
    if rank == 0:       # the whole batch, eagerly, on one rank at the initial parameters (a batch-mean loss: its gradient is the MEAN of the ranks')
        whole = build_visual_bert(cfg, sd); whole.eval()
        out = whole(SampleList(sample_to(full, "cuda")))
        sum(v.sum() for v in out["losses"].values()).backward()
        ref = {n: (p.grad.detach().float().cpu().numpy() if p.grad is not None else None) for n, p in whole.named_parameters()}
        ref["__init__"] = {n: p.detach().float().cpu().numpy() for n, p in whole.named_parameters()}
    q.put((rank, got, ref, wires))
    D.synchronize()
    dist.destroy_process_group()


def test_two_ranks_chained_graphs_apply_the_whole_batch_update():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            rank, got, ref, wires = q.get(timeout=150)
            res[rank] = (got, ref, wires)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    # bf16 wire, fp32 wire for the small embedding tables, and (round 5) the word-embedding gradient as touched rows: every rank sends the rows
    # its batch touched, every rank rebuilds the sum in the same order — the `torch.equal(g0, g1)` / `torch.equal(p0, p1)` below include that table
    assert res[0][2] == ["torch.bfloat16", "torch.float32", "rows:model.bert.embeddings.word_embeddings.weight"], res[0][2]
    ref = res[0][1]
    init = ref.pop("__init__")
    g0s, g1s = res[0][0].pop("__grads__"), res[1][0].pop("__grads__")
    checked = 0
    for name, r in ref.items():
        p0, p1 = torch.from_numpy(res[0][0][name]), torch.from_numpy(res[1][0][name])
        assert torch.equal(p0, p1), name                   # replicas stay identical
        if r is None:
            assert g0s[name] is None and g1s[name] is None, name
            assert torch.equal(p0, torch.from_numpy(init[name])), name          # never-used parameters are not touched
            continue
        g0, g1, r = torch.from_numpy(g0s[name]), torch.from_numpy(g1s[name]), torch.from_numpy(r)
        assert torch.equal(g0, g1), name
        g0 = g0 * 0.5                                      # the buffers hold the SUM; the update applies optimizer.grad_scale = 1 / world
        scale = float(r.abs().max()) + 1e-12
        tol = 3e-2 if "embeddings" not in name else 1e-2    # bf16 on the wire (fp32 for the embedding tables) + bf16 kernels at B = 2 vs 4
        assert float((g0 - r).abs().max()) <= tol * scale, (name, float((g0 - r).abs().max()), scale)
        assert not torch.equal(p0, torch.from_numpy(init[name])), name          # ... and two optimizer steps moved the parameter
        checked += 1
    assert checked > 30


def _rccl_worker(port, q):
    """One rank, the REAL backend ("nccl" = RCCL): the chained step with its all-reduce calls issued between the stage graphs (a 1-rank
    communicator makes every sum the identity) against the same step without any collective."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.pop("MMF_AMD_DIST_BACKEND", None)
    import torch.distributed as dist
    from mmf_amd.common.sample import SampleList
    from mmf_amd.utils.graph import GraphedDataParallelStep
    from tests.golden_utils import load_case
    from tests.model_utils import build_visual_bert, sample_to
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.ones(4096, device="cuda", dtype=torch.bfloat16)
    dist.all_reduce(t, async_op=True).wait()
    torch.cuda.synchronize()
    assert float(t.float().sum()) == 4096.0
    z, case, cfg, sd, sample = load_case("small64")
    batch = SampleList(sample_to(sample, "cuda"))
    out = {}
    for forced in (False, True):
        model = build_visual_bert(cfg, sd); model.eval()
        opt = _optimizer(model, True)
        step = GraphedDataParallelStep(model, batch, _layers(model), opt, warmup=1, comm_dtype=torch.bfloat16)
        if forced:
            step.world = 2              # issue the collectives (bf16 and fp32 wire buffers, async, waited for one stage later)
            opt.grad_scale = 1.0        # ... whose sum over the ONE rank is the identity
        losses = [float(step()) for _ in range(3)]
        torch.cuda.synchronize()
        out[forced] = (losses, {n: p.detach().float().cpu().numpy() for n, p in model.named_parameters()})
    q.put(out)
    dist.destroy_process_group()


def test_chained_graphs_with_rccl_collectives_between_the_stages():
    """The N > 1 launch structure with the real communicator library: RCCL's all_reduce (bf16 and fp32 buffers, async_op, wait one stage later)
    interleaved with hipGraph replays on one rank must leave losses and parameters bit-identical to the chain without collectives."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q), daemon=True)
    p.start()
    try:
        out = q.get(timeout=200)
        p.join(timeout=60)
        assert p.exitcode == 0
    finally:
        if p.is_alive():
            p.kill()
    (l0, p0), (l1, p1) = out[False], out[True]
    assert l0 == l1, (l0, l1)
    assert l0[2] < l0[0]
    for n in p0:
        assert (p0[n] == p1[n]).all(), n


def test_touched_row_exchange_one_rank_is_bit_identical_to_the_dense_path():
    """`sparse_rows=True` on one rank runs the whole touched-row path of the word-embedding gradient (duplicate ids marked, rows gathered from the
    dense gradient in the backward stage graph, stable sort + segment sum ahead of AdamW in the update graph): the sums are the rows themselves, so
    losses and every parameter must equal the dense path bit for bit; the table is picked automatically (the other tables are too small, and
    a table with a second kind of contribution would be refused)."""
    from mmf_amd.common.sample import SampleList
    from mmf_amd.utils.graph import GraphedDataParallelStep
    from tests.golden_utils import load_case
    from tests.model_utils import build_visual_bert, sample_to
    z, case, cfg, sd, sample = load_case("small64")
    batch = SampleList(sample_to(sample, "cuda"))
    out = {}
    for sparse in (False, True):
        model = build_visual_bert(cfg, sd); model.eval()
        opt = _optimizer(model, True)
        step = GraphedDataParallelStep(model, batch, _layers(model), opt, warmup=1, sparse_rows=sparse)
        names = {id(p): n for n, p in model.named_parameters()}
        assert [names[id(sp.p)] for sp in step.sparse] == (["model.bert.embeddings.word_embeddings.weight"] if sparse else [])
        losses = [float(step()) for _ in range(3)]
        torch.cuda.synchronize()
        out[sparse] = (losses, {n: p.detach().clone() for n, p in model.named_parameters()})
    assert out[False][0] == out[True][0]
    for n in out[False][1]:
        assert torch.equal(out[False][1][n], out[True][1][n]), n


def test_segment_sum_rows_is_deterministic_and_matches_index_add():
    from mmf_amd import _native as nat
    g = torch.Generator().manual_seed(9)
    V, H, M = 500, 96, 4096
    ids = torch.randint(-1, 60, (M,), generator=g)                 # many collisions, some -1 (removed duplicates)
    rows = torch.randn(M, H, generator=g)
    sorted_ids, perm = ids.cuda().sort(stable=True)
    out = torch.full((V, H), 3.0, device="cuda")
    nat.segment_sum_rows_f32(sorted_ids, perm, rows.cuda(), out)
    ref = torch.full((V, H), 3.0, dtype=torch.float64)
    touched = sorted({int(i) for i in ids.tolist() if i >= 0})
    ref[touched] = 0.0
    keep = ids >= 0
    ref.index_add_(0, ids[keep], rows[keep].double())
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=1e-5)
    again = torch.full((V, H), 3.0, device="cuda")
    nat.segment_sum_rows_f32(sorted_ids, perm, rows.cuda(), again)
    assert torch.equal(out, again)
    # fixed order: the sum of a segment is the left-to-right fp32 sum in stable-sorted order
    seq = torch.zeros(H)
    for j in torch.nonzero(ids == touched[0]).flatten().tolist():
        seq = seq + rows[j]
    assert torch.equal(out[touched[0]].cpu(), seq)
