"""hipGraph capture of the training step.

The step is ~450 short kernels; launched one by one from Python the host (ctypes + autograd bookkeeping)
becomes the bottleneck once the kernels are fast.  `GraphedTrainStep` runs a few eager warm-up steps,
then captures zero-grad-free forward + loss + backward into ONE hipGraph and replays it: no Python, no
per-kernel launch cost.  Everything the path launches is capture-safe by construction (only stream-ordered
work, no host sync, no allocation outside torch's caching allocator).

Dropout under replay: site keys are baked into the captured kernels, so a device word (`self.seed`) is mixed
into every key at run time and `mmf_seed_advance` — the first node of the graph — bumps it, i.e. each replay
draws fresh masks while forward and backward of the same replay agree (mmf_amd.functional._DropoutKeys).
The fp32 -> bf16 weight-shadow casts are captured too, so replays see parameter updates made between them.
"""
import gc
import os

import torch
import torch.distributed as dist

from mmf_amd import functional as Fn
from mmf_amd.common.registry import registry
from mmf_amd.common.sample import SampleList


def release_autograd_state():
    """Drop what keeps an earlier eager step's autograd graph alive on our side (the losses `Losses.forward`
    registers, losses.py:127-131) and collect garbage.

    Why it matters: a parameter's AccumulateGrad node is bound to the stream it was first created on and lives as
    long as any graph references it.  If an eager forward ran on the legacy default stream and its outputs are still
    alive, the captured backward hands gradients to those nodes, the default stream is dragged into the capture and
    `hipStreamEndCapture` crashes.  Callers must also drop their own references (`del out, loss`) — or simply run
    eager steps under `torch.cuda.stream(...)`."""
    registry.unregister("losses")
    gc.collect()


def _clone_batch(batch):
    out = SampleList()
    for k in batch.fields():
        v = batch[k]
        if isinstance(v, torch.Tensor):
            out[k] = v.clone()
        elif isinstance(v, SampleList):
            out[k] = _clone_batch(v)
        else:
            out[k] = v
    return out


def _copy_batch(dst, src):
    for k in dst.fields():
        v = dst[k]
        if isinstance(v, torch.Tensor):
            v.copy_(src[k], non_blocking=True)
        elif isinstance(v, SampleList):
            _copy_batch(v, src[k])


def total_loss(out):
    """The scalar the trainer differentiates: the sum over the loss dict of each entry's sum (mmf/trainers/core/training_loop.py:199-213 sums the
    entries' means; every loss of this package is already a scalar).  A single scalar entry is returned as it is: `tensor.sum()` of a 0-dim tensor
    and Python's `0 + tensor` are two more kernels (a reduction and an add) that compute nothing."""
    vals = list(out["losses"].values())
    if len(vals) == 1 and vals[0].numel() == 1:      # (MMFLoss hands scalars on as shape [1], like the reference: losses.py `loss.view(1)`)
        return vals[0].reshape(())
    return sum(v.sum() for v in vals)


class GraphedTrainStep:
    """`optimizer` (an `adam_w` built with `capturable=True`) puts the parameter update into the graph as well: one
    replay = one full training step.  The optimizer then also keeps the bf16 weight shadows current, so no cast kernel
    is captured; without it the casts are captured so that replays see updates made between them.

    The eager warm-up passes (they populate the allocator and the weight shadows) run forward + backward only: no
    parameter is updated and no step is counted before the first replay; the optimizer's moments are allocated up front
    (`ensure_state`) so that the captured update contains no zero-fill."""

    def __init__(self, model, batch, warmup=3, loss_of=None, optimizer=None, seed=None):
        # (Rounds 2 - 3 could also put the grouped weight gradients or each layer's AdamW update on a second stream beside the backward of the layers
        #  below; both measured slower on every box - the streamed 213 MB per layer evict the GEMMs' operand panels from L2, 8.53 against 8.25 ms -
        #  and were removed in round 6: profiles/r04_experiments.txt.)
        self.model = model
        self.optimizer = optimizer
        if optimizer is not None and not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs an optimizer whose step reads its counters from device memory (capturable=True)")
        self.loss_of = loss_of or total_loss
        release_autograd_state()
        self.static_batch = _clone_batch(batch)
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = next(model.parameters()).device
        self.seed = torch.zeros(1, dtype=torch.int32, device=dev) if seed is None else seed      # (shared between the steps of a BucketedTrainStep)
        if optimizer is not None:
            optimizer.ensure_state(self.params)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), Fn.dropout_keys.graph_mode(self.seed):
            for _ in range(warmup):
                self._eager(update=False)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        model.zero_grad(set_to_none=True)
        if optimizer is None:
            Fn.shadows.clear()   # so the weight-shadow casts are part of the captured step
        with Fn.dropout_keys.graph_mode(self.seed):
            with torch.cuda.graph(self.graph):
                self.out, self.loss = self._eager()

    def _eager(self, update=True):
        return self._eager_body(update)

    def _eager_body(self, update):
        # the dropout seed word and the optimizer's step count / schedule factor advance in ONE one-thread launch at the head of the step
        head_advance = self.optimizer is not None and update and hasattr(self.optimizer, "advance")
        if head_advance:
            self.optimizer.advance(self.seed)
        else:
            Fn.nat.seed_advance(self.seed)
        out = self.model(self.static_batch)
        loss = self.loss_of(out)
        # torch.autograd.grad instead of loss.backward(): AccumulateGrad nodes are bound to the stream they were first
        # created on; if an earlier eager step created them on the legacy default stream, running them inside the
        # capture drags that stream into it and hipStreamEndCapture crashes.  Capturing the gradients directly keeps
        # every captured node on the capture stream.
        with Fn.ln_defer(), Fn.wgrad_defer():
            # (the seed of the backward pass is a tensor made once, ahead of the capture: `grad_outputs=None` fills a new ones tensor per step)
            if getattr(self, "_one", None) is None or self._one.dtype != loss.dtype:
                self._one = torch.ones_like(loss)
            grads = torch.autograd.grad(loss, self.params, grad_outputs=self._one, allow_unused=True)
        self.grads = grads
        for p, g in zip(self.params, grads):
            p.grad = g
        if self.optimizer is not None and update:
            self.optimizer.step(**({"advance": False} if head_advance else {}))
        return out, loss

    def bind_gradients(self):
        """Point every `p.grad` at THIS step's captured gradient tensors (after another step captured or replayed over the same parameters)."""
        for p, g in zip(self.params, self.grads):
            p.grad = g

    def __call__(self, batch=None):
        if batch is not None:
            _copy_batch(self.static_batch, batch)
        self.graph.replay()
        return self.loss


def _batch_signature(batch):
    sig = []
    for k in batch.fields():
        v = batch[k]
        if isinstance(v, torch.Tensor):
            sig.append((k, tuple(v.shape), str(v.dtype)))
        elif isinstance(v, SampleList):
            sig.append((k, _batch_signature(v)))
    return tuple(sig)


class BucketedTrainStep:
    """The graphed training step for batches whose TEXT LENGTH varies: one `GraphedTrainStep` per batch shape, captured the first time the shape is
    seen (a few eager warm-up passes without update, then the capture), replayed afterwards.  With `trim=m` (> 0) the step first cuts the text
    columns no sample of the batch uses, rounded up to a multiple of m (`mmf_amd.common.prefetch.trim_text_padding`: results are those of the
    untrimmed batch) — for a batch already in HBM that reads the mask back (one synchronisation per step); feed it from
    `DevicePrefetcher(trim_text_padding=m)` with `trim=0` here and nothing waits.  All buckets share the model, the optimizer (its moments, step count
    and schedule words live in device memory the captured updates read), the weight shadows and the dropout seed word; each owns its activations
    (a private pool of ~ 70 MB per sample position block: 2.2 GB at 228 positions, B = 32 — 288 GB of HBM hold every bucket of a 128-token model)."""

    def __init__(self, model, optimizer=None, warmup=2, loss_of=None, trim=8, max_buckets=32):
        self.model, self.optimizer, self.warmup, self.loss_of = model, optimizer, warmup, loss_of
        self.trim, self.max_buckets = int(trim or 0), max_buckets
        self.steps = {}
        self.seed = torch.zeros(1, dtype=torch.int32, device=next(model.parameters()).device)
        self._bound = None

    def __call__(self, batch):
        if self.trim > 0:
            from mmf_amd.common.prefetch import trim_text_padding
            batch = trim_text_padding(batch, self.trim)
        key = _batch_signature(batch)
        step = self.steps.get(key)
        if step is None:
            if len(self.steps) >= self.max_buckets:
                raise RuntimeError("BucketedTrainStep: more than %d batch shapes; raise `trim` (coarser buckets) or max_buckets" % self.max_buckets)
            step = self.steps[key] = GraphedTrainStep(self.model, batch, warmup=self.warmup, loss_of=self.loss_of, optimizer=self.optimizer, seed=self.seed)
            self._bound = step          # (the capture left p.grad on its own tensors)
        loss = step(batch)
        if self._bound is not step:
            step.bind_gradients()
            self._bound = step
        return loss


_SPARSE_CHECK = int(os.environ.get("MMF_AMD_SPARSE_CHECK", "0") or 0)      # k > 0: every k-th step verifies the touched-row contract (a host read-back)


class _SparseRows:
    """Touched-row exchange of ONE embedding-table gradient between data-parallel ranks: the reference's DDP all-reduces the dense
    [30522, 768] fp32 word-embedding gradient (97 MB, the last gradient backward produces: the un-overlappable tail of the step,
    mmf/trainers/core/device.py:104-110) although a rank's batch touches at most B * T = 4096 of its rows.  Here every rank sends its ids and its
    rows (12.6 MB), all-gathered, and rebuilds the SUM itself:

      backward stage graph   ids sorted, duplicates marked -1 (the local dense gradient already holds their sum in ONE row), rows gathered
                             from the dense gradient into the wire buffer                                              [`pack`]
      between the graphs     all_gather of ids and rows (RCCL; a zero-padded all_reduce over gloo, which has no CUDA all_gather)      [`exchange`]
      update stage graph     STABLE sort of the gathered ids, then `mmf_segment_sum_rows_f32`: one wave per id adds that id's rows in sorted
                             (= rank) order without atomics and writes the sum into the dense gradient the fused AdamW reads        [`merge`]

    Every rank computes the same function of the same gathered data in the same order: the replicas stay bit-identical (tested with two ranks).
    Rows no rank touched keep the zeros of the local dense gradient; rows touched elsewhere only are written by the merge."""

    def __init__(self, p, stage, n_ids, world, rank, group, dev):
        self.p, self.stage, self.n, self.world, self.rank, self.group = p, stage, int(n_ids), world, rank, group
        H = p.shape[1]
        self.ids_wire = torch.full((self.n,), -1, dtype=torch.int64, device=dev)
        self.rows_wire = torch.zeros(self.n, H, dtype=torch.float32, device=dev)
        one = world <= 1
        self.ids_all = self.ids_wire if one else torch.full((world * self.n,), -1, dtype=torch.int64, device=dev)
        self.rows_all = self.rows_wire if one else torch.zeros(world * self.n, H, dtype=torch.float32, device=dev)
        self.dense = None
        self.nccl = (not one) and dist.get_backend(group) == "nccl"

    def pack(self, g, ids):          # captured in the backward stage's graph
        flat = ids.reshape(-1)
        sorted_ids, _ = flat.sort()
        dup = torch.zeros_like(sorted_ids, dtype=torch.bool)
        dup[1:] = sorted_ids[1:] == sorted_ids[:-1]
        self.ids_wire.copy_(torch.where(dup, torch.full_like(sorted_ids, -1), sorted_ids))
        Fn.nat.gather_rows2_f32(g, g, self.ids_wire, self.rows_wire, self.n, g.shape[1])      # (-1 reads row 0: the receivers skip the entry)
        self.dense = g               # the stage graph's own gradient tensor (kept alive by the step): the merge rewrites its touched rows in place

    def exchange(self):              # eager, between the graph replays
        if self.world <= 1:
            return []
        if self.nccl:
            return [dist.all_gather_into_tensor(self.ids_all, self.ids_wire, group=self.group, async_op=True),
                    dist.all_gather_into_tensor(self.rows_all, self.rows_wire, group=self.group, async_op=True)]
        # gloo (tests: two ranks sharing one GPU) has no all_gather for device tensors: every rank fills its slot of a zero buffer, summed
        self.ids_all.zero_()
        self.ids_all.view(self.world, self.n)[self.rank].copy_(self.ids_wire + 1)
        self.rows_all.zero_()
        self.rows_all.view(self.world, self.n, -1)[self.rank].copy_(self.rows_wire)
        dist.all_reduce(self.ids_all, group=self.group)
        dist.all_reduce(self.rows_all, group=self.group)
        self.ids_all.sub_(1)
        return []

    def verify(self):                # debugging (MMF_AMD_SPARSE_CHECK=k: every k-th step), eager, right after the backward stage's replay
        """The contract of the touched-row exchange: the stage's dense gradient is zero outside the rows its packed ids name.  The choice was made
        once, from the eager warm-up batch (`_pick_sparse`); a contribution to another row - ids passed by keyword past the hook, a second lookup
        the warm-up batch happened to cover - would be dropped on the OTHER ranks without any error.  This check reads back and raises."""
        ids = self.ids_wire[self.ids_wire >= 0]
        touched = torch.zeros(self.dense.shape[0], dtype=torch.bool, device=self.dense.device)
        touched[ids] = True
        if bool((self.dense[~touched] != 0).any()):
            raise RuntimeError("touched-row exchange: the gradient of a [%d, %d] table has non-zero rows outside the ids its embedding stage looked up; "
                               "run with sparse_rows=False" % tuple(self.dense.shape))

    def merge(self):                 # captured in the update stage's graph, ahead of the fused AdamW
        sorted_all, perm = self.ids_all.sort(stable=True)
        Fn.nat.segment_sum_rows_f32(sorted_all, perm, self.rows_all, self.dense)


class GraphedDataParallelStep:
    """The training step of ONE data-parallel rank as a short chain of hipGraphs with the gradient all-reduces between them
    (collective C1 of SURVEY.md section 2.3; the reference wraps the model in DistributedDataParallel, mmf/trainers/core/device.py:104-110):

        [F B_0] | [B_1] | [O_0 B_2] | [O_1 B_3] | ... | [O_(n-2) O_(n-1)]         (F: forward + loss; B_j: backward stage j, cut at the outputs of
           `-> all-reduce of stage 0's gradients on the communicator's stream,      `cuts`; O_j: AdamW of the parameters of stage j; [..]: ONE hipGraph)
               while [B_1] replays; [O_0 B_2] waits for it, ...

    so the host enqueues n + 1 graph launches and the collectives per step instead of ~450 kernels (the eager N > 1 step is host-bound: bench.py's
    "eager" leg) and the collectives themselves stay outside the graphs (RCCL launched eagerly between replays — nothing
    depends on collective-in-graph support).  `cuts` are modules whose output tensor splits the network (e.g. three encoder
    layers): the forward hands the next module a detached copy, which makes each segment its own autograd graph; stage j of
    the backward is `torch.autograd.grad` of segment n - j, seeded with the gradient the previous stage produced for the
    detached copy.  Which parameters belong to which stage is discovered once in the eager warm-up (a parameter used in
    several segments is summed and travels with the last stage that touches it; parameters that never receive a gradient —
    the BertPooler under `pooler_strategy: vqa` — are left with `grad = None`, the same set on every rank).

    Gradients of a stage are packed by the stage's graph into one flat buffer per wire type — `comm_dtype` (bf16: half the
    xGMI bytes) and fp32 for `fp32_params` (default: embedding tables, whose rows collect sparse, differently scaled
    contributions) — and summed over the ranks IN PLACE; the optimizer reads the summed wire buffers directly
    (`AdamW.external_grads`: the fused update converts bf16 gradients while it loads them — round 2 unpacked them into a second
    fp32 buffer first, 0.7 GB of extra traffic per step), and the 1 / world_size of the mean is folded into the update
    (`optimizer.grad_scale`).  The update itself is cut per stage: `O_j` replays as soon as stage j's all-reduce has landed, after
    the NEXT backward stage has been enqueued, so the updates of the early stages run while the later collectives are still on the
    wire and the only exposed tail is the last stage's collective (the fp32 word-embedding bucket) plus its own update.
    `gradients()` returns the reduced buffers per parameter (`.grad` stays unset: a bf16 buffer cannot be an fp32 parameter's grad).
    The optimizer must be `capturable=True`; its state is allocated before the capture (`ensure_state`) and the warm-up runs no
    optimizer step: the first replay is step 1."""

    def __init__(self, model, batch, cuts, optimizer, process_group=None, comm_dtype=None, fp32_params=None, warmup=2, loss_of=None,
                 sparse_rows=None):
        """`sparse_rows`: exchange only the TOUCHED rows of embedding tables whose gradient is row-sparse (the word embeddings: <= B * T of 30522
        rows receive a gradient) instead of all-reducing the dense table — see `_SparseRows`.  None = on when world > 1 (MMF_AMD_SPARSE_ROWS=0
        switches it off), True forces it (a one-rank run then exercises the whole path), False = never."""
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedDataParallelStep needs an optimizer whose step reads its counters from device memory (capturable=True)")
        self.model, self.optimizer, self.group = model, optimizer, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        if comm_dtype is None:
            comm_dtype = torch.bfloat16 if self.world > 1 else torch.float32
        self.comm_dtype = comm_dtype
        self.loss_of = loss_of or total_loss
        self.cuts = list(cuts)
        release_autograd_state()
        self.static_batch = _clone_batch(batch)
        self.params = [p for p in model.parameters() if p.requires_grad]
        if fp32_params is None:
            fp32_ids = {id(m.weight) for m in model.modules() if isinstance(m, torch.nn.Embedding)}
        else:
            fp32_ids = {id(p) for p in fp32_params}
        dev = self.params[0].device
        self.seed = torch.full((1,), 7919 * rank, dtype=torch.int32, device=dev)      # ranks draw different dropout masks
        self._bounds, self._hooking = [], False
        if sparse_rows is None:
            sparse_rows = self.world > 1 and os.environ.get("MMF_AMD_SPARSE_ROWS", "1") != "0"
        self._sparse_on, self._ids, self._probe, self.sparse, self._rank = bool(sparse_rows), {}, {}, [], rank
        handles = [m.register_forward_hook(self._cut_hook) for m in self.cuts]
        if self._sparse_on:      # the ids every embedding stage looks up in its word table (the module's first positional argument)
            handles += [m.register_forward_pre_hook(self._ids_hook) for m in model.modules()
                        if isinstance(getattr(m, "word_embeddings", None), torch.nn.Embedding)]
        try:
            optimizer.ensure_state(self.params)
            optimizer.grad_scale = 1.0 / self.world
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), Fn.dropout_keys.graph_mode(self.seed):
                # eager warm-up (no optimizer step); the first pass also assigns parameters to backward stages
                self.stage_params = None
                for _ in range(max(1, warmup)):
                    self._forward()
                    self._discover_or_run()
                    self._bounds = []
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            model.zero_grad(set_to_none=True)
            release_autograd_state()
            self._pick_sparse(dev)
            self._probe = {}
            self._layout(fp32_ids, dev)
            # capture_error_mode="thread_local": the communicator's watchdog thread polls events while we capture
            pool = torch.cuda.graph_pool_handle()
            # Round 6: FIVE graph launches per step for four stages instead of nine.  The forward rides with backward stage 0 (no collective lies
            # between them), the update of stage j - 2 rides at the head of backward stage j (that is where it sat in stream order already: behind
            # stage j - 1, whose replay covers all-reduce j - 2), the last two updates share the tail graph:
            #     [F B_0] ar_0^ [B_1] ar_1^ (ar_0) [O_0 B_2] ar_2^ (ar_1) [O_1 B_3] ar_3^ (ar_2, ar_3) [O_2 O_3]
            # A graph launch costs ~12 us of host time and leaves a ~40 us hole on the GPU: the chain at N = 1 ran 0.44 ms behind the single graph.
            n = len(self.stage_params)
            self.g_bwd = [torch.cuda.CUDAGraph() for _ in range(n)]       # graph k holds backward stage k (k = 0: the forward too; k >= 2: update k - 2 first)
            self.g_tail = torch.cuda.CUDAGraph()                          # updates n - 2 and n - 1

            def stage_ids(j):
                return {id(p) for p in self.buckets[j]["p16"]} | {id(p) for p in self.buckets[j]["p32"]} | {id(sp.p) for sp in self.sparse if sp.stage == j}

            first_update = [True]

            def update(j):       # captured: the ranks' touched rows -> the summed dense gradient, then the fused AdamW of stage j's parameters
                ids = stage_ids(j)
                if not ids and j > 0:
                    return           # (a stage without parameters of its own: nothing to update)
                optimizer.external_grads = self._wire_views()
                for sp in self.sparse:
                    if sp.stage == j:
                        sp.merge()
                optimizer.step(only=ids, advance=first_update[0])      # only the first update of a step advances the step count / schedule
                first_update[0] = False

            with Fn.dropout_keys.graph_mode(self.seed):
                carry = None
                self._keep = []
                for j, g in enumerate(self.g_bwd):
                    with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                        if j == 0:
                            self.out, self.loss = self._forward()
                        if j >= 2:
                            update(j - 2)
                        grads, carry = self._stage_grads(j, carry, self.stage_params[j])
                        self._pack(j, grads)
                    self._keep.append((grads, carry))
                with torch.cuda.graph(self.g_tail, pool=pool, capture_error_mode="thread_local"):
                    if n >= 2:
                        update(n - 2)
                    update(n - 1)
        finally:
            for h in handles:
                h.remove()
        self._bounds = []

    # ---- forward with cut points --------------------------------------------------------------------------------------
    def _cut_hook(self, module, inputs, output):
        if not self._hooking:
            return None
        t = output[0] if isinstance(output, tuple) else output
        d = t.detach().requires_grad_()
        self._bounds.append((t, d))
        return (d,) + tuple(output[1:]) if isinstance(output, tuple) else d

    def _ids_hook(self, module, args):
        if self._hooking and args and isinstance(args[0], torch.Tensor) and args[0].dtype == torch.int64:
            k = id(module.word_embeddings.weight)
            self._ids[k] = args[0] if k not in self._ids else None      # (a table looked up by TWO embedding stages in one forward is not a candidate)

    def _forward(self):
        self._ids = {}
        Fn.nat.seed_advance(self.seed)
        self._bounds, self._hooking = [], True
        try:
            out = self.model(self.static_batch)
        finally:
            self._hooking = False
        if len(self._bounds) != len(self.cuts):
            raise RuntimeError("GraphedDataParallelStep: %d of the %d cut modules ran in the forward" % (len(self._bounds), len(self.cuts)))
        self._loss_t = self.loss_of(out)
        return out, self._loss_t

    # ---- backward, one stage at a time ------------------------------------------------------------------------------------
    def _stage_grads(self, j, carry, params):
        """Stage j (0 = the segment that ends in the loss): gradients of `params` and of the detached input of the segment."""
        n = len(self._bounds)
        root = self._loss_t if j == 0 else self._bounds[n - j][0]
        inputs = list(params) + ([self._bounds[n - j - 1][1]] if j < n else [])
        with Fn.ln_defer(), Fn.wgrad_defer():        # the stage's LayerNorm parameter gradients / queued weight gradients are finished at its end
            if j == 0 and (getattr(self, "_one", None) is None or self._one.dtype != root.dtype):
                self._one = torch.ones_like(root)
            grads = torch.autograd.grad(root, inputs, grad_outputs=self._one if j == 0 else carry, allow_unused=True)
        if j < n:
            if grads[-1] is None:
                raise RuntimeError("GraphedDataParallelStep: the loss does not depend on the output of cut %d" % (n - j - 1))
            return list(grads[:-1]), grads[-1]
        return list(grads), None

    def _discover_or_run(self):
        first = self.stage_params is None
        stages, carry = [], None
        self._probe = {}
        for j in range(len(self.cuts) + 1):
            cand = self.params if first else self.stage_params[j]
            grads, carry = self._stage_grads(j, carry, cand)
            stages.append([p for p, g in zip(cand, grads) if g is not None])
            if self._sparse_on:
                for p, g in zip(cand, grads):
                    if g is not None and self._ids.get(id(p)) is not None:
                        self._probe.setdefault(id(p), []).append((j, g))
        if first:
            self.stage_params = stages
            seen = {}
            for j, plist in enumerate(stages):
                for p in plist:
                    seen.setdefault(id(p), []).append(j)
            self._shared = {k: v for k, v in seen.items() if len(v) > 1}     # id -> stages; packed with the last one

    # ---- flat wire buffers ------------------------------------------------------------------------------------------------
    def _layout(self, fp32_ids, dev):
        self.buckets = []        # per stage: dict(p16, o16, wire16, p32, o32, wire32)
        last_stage = {k: v[-1] for k, v in self._shared.items()}
        self._partial = {k: None for k in self._shared}      # running sums of shared parameters
        for j, plist in enumerate(self.stage_params):
            mine = [p for p in plist if last_stage.get(id(p), j) == j]
            b = dict(p16=[], o16=[], p32=[], o32=[])
            n16 = n32 = 0
            for p in mine:
                if any(sp.p is p for sp in self.sparse):
                    continue             # exchanged as touched rows (_SparseRows), not inside a wire buffer
                wide = id(p) in fp32_ids or self.comm_dtype == torch.float32
                (b["p32"] if wide else b["p16"]).append(p)
                if wide:
                    b["o32"].append(n32); n32 += (p.numel() + 63) // 64 * 64
                else:
                    b["o16"].append(n16); n16 += (p.numel() + 63) // 64 * 64
            b["wire16"] = torch.zeros(n16, dtype=self.comm_dtype, device=dev) if n16 else None
            b["wire32"] = torch.zeros(n32, dtype=torch.float32, device=dev) if n32 else None
            self.buckets.append(b)

    def _pack(self, j, grads):
        b = self.buckets[j]
        have = {}
        for p, g in zip(self.stage_params[j], grads):
            if id(p) in self._shared:
                acc = self._partial[id(p)]
                g = g if acc is None else acc + g
                self._partial[id(p)] = g
            have[id(p)] = g
        for sp in self.sparse:
            if sp.stage == j:
                sp.pack(have[id(sp.p)], self._ids[id(sp.p)])
        for plist, offs, flat in ((b["p16"], b["o16"], b["wire16"]), (b["p32"], b["o32"], b["wire32"])):
            if plist:
                srcs = [have[id(p)] for p in plist]
                if all(g.dtype == torch.float32 and g.is_contiguous() for g in srcs):
                    # ONE multi-tensor HIP launch per wire buffer: every gradient read once, converted to the wire type, written once (round 4:
                    # `_foreach_copy_`, one converting kernel per gradient inside the stage graph)
                    Fn.nat.pack_f32_multi(srcs, offs, flat, 1.0)
                else:
                    torch._foreach_copy_([flat[o:o + p.numel()].view_as(p) for p, o in zip(plist, offs)], srcs)

    def _wire_views(self):
        """id(parameter) -> its slice of the stage's wire buffer (after the all-reduce: the SUM over the ranks), in the wire's dtype."""
        views = {}
        for b in self.buckets:
            for plist, offs, flat in ((b["p16"], b["o16"], b["wire16"]), (b["p32"], b["o32"], b["wire32"])):
                for p, o in zip(plist, offs):
                    views[id(p)] = flat[o:o + p.numel()].view_as(p)
        for sp in self.sparse:
            views[id(sp.p)] = sp.dense
        return views

    def _pick_sparse(self, dev):
        """Which parameters travel as touched rows: embedding tables whose ONLY gradient contributions in the warm-up were the rows of the ids
        their embedding stage looked up (a decoder tied to the table, a second lookup of other ids — MMBT's start / end tokens — make the gradient
        dense or name other rows: those stay in the wire buffers), with at least four table rows per looked-up id, living in ONE backward stage.
        The decision is agreed over the ranks (MIN), so every rank issues the same collectives."""
        if not self._sparse_on:
            return
        cands = []
        for p in self.params:
            ent, ids = self._probe.get(id(p)), self._ids.get(id(p))
            ok = ent is not None and len(ent) == 1 and ids is not None and p.dim() == 2 and p.shape[0] >= 4 * ids.numel() and p.shape[1] % 4 == 0
            if ok:
                j, g = ent[0]
                touched = torch.zeros(p.shape[0], dtype=torch.bool, device=dev)
                touched[ids.reshape(-1).clamp(0, p.shape[0] - 1)] = True
                ok = g.dtype == torch.float32 and not bool((g[~touched] != 0).any())      # (a host read-back: this is the eager warm-up)
            cands.append((p, ok, ent[0][0] if ent else -1))
        flags = torch.tensor([1 if ok else 0 for _, ok, _ in cands], dtype=torch.int32, device=dev)
        if self.world > 1 and flags.numel():
            dist.all_reduce(flags, op=dist.ReduceOp.MIN, group=self.group)
        for (p, _, j), f in zip(cands, flags.tolist()):
            if f:
                self.sparse.append(_SparseRows(p, j, self._ids[id(p)].numel(), self.world, self._rank, self.group, dev))

    def gradients(self):
        """After a step: parameter -> reduced gradient buffer (the sum over the ranks; multiply by `optimizer.grad_scale` for the mean)."""
        views = self._wire_views()
        return {p: views[id(p)] for p in self.params if id(p) in views}

    # ---- one training step ------------------------------------------------------------------------------------------------
    def __call__(self, batch=None):
        if batch is not None:
            _copy_batch(self.static_batch, batch)
        self._calls = getattr(self, "_calls", 0) + 1
        check = _SPARSE_CHECK > 0 and self._calls % _SPARSE_CHECK == 0
        pending = []         # pending[j]: the collectives of backward stage j
        for j, g in enumerate(self.g_bwd):
            if j >= 2:       # graph j opens with the update of stage j - 2: its gradients must have been summed
                for w in pending[j - 2]:
                    w.wait()
            g.replay()
            if check:
                for sp in self.sparse:
                    if sp.stage == j:
                        sp.verify()
            works = []
            if self.world > 1:
                b = self.buckets[j]
                for flat in (b["wire16"], b["wire32"]):
                    if flat is not None:
                        works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                for sp in self.sparse:
                    if sp.stage == j:
                        works += sp.exchange()
            pending.append(works)
        for ws in pending[-2:]:
            for w in ws:
                w.wait()
        self.g_tail.replay()
        return self.loss
