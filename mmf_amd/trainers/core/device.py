"""Data-parallel gradient synchronisation for one node of MI355Xs.

Replaces the `DistributedDataParallel` wrap at mmf/trainers/core/device.py:104-110 (collective C1 of
SURVEY.md §2.3: a mean all-reduce of all 114 M gradients every step).  Same semantics — every rank
ends `backward()` with the average gradient in `param.grad`, parameters that received no gradient on
ANY rank (the BertPooler under `pooler_strategy: vqa`; the reference needs `find_unused_parameters=True`
for it, tools/sweeps/sweep_visual_bert.py:41) keep `grad = None` so the optimizer skips them — but shaped
for xGMI:

  * gradients are packed into a few LARGE flat buckets (default 64 MiB of fp32) in the order backward
    produces them (head -> layer 11 ... 0 -> embeddings), because an xGMI ring all-reduce is bound by one
    153 GB/s link and per-collective latency, not by switch bandwidth;
  * the bucket LAYOUT is fixed at construction (one 256-byte-aligned slot per parameter, zero-filled when the
    parameter has no gradient on this rank), so every rank always issues collectives of identical size
    whatever its local used-parameter set is;
  * buckets travel as fp32 by default — the reference's DDP all-reduce is fp32 and this class is its drop-in —
    and as bf16 when asked to (`comm_dtype=torch.bfloat16`: half the xGMI bytes, 229 MB instead of 458 MB per
    step; each rank's contribution is scaled by 1 / world BEFORE it is rounded, so the bf16 sum cannot overflow
    and the wire carries the mean; a deviation from the reference's arithmetic held to the bf16 bound in the
    tests).  With a bf16 wire, parameters whose gradient wants fp32 — by default embedding tables, whose rows
    receive sparse, differently scaled contributions — go into buckets of their own that stay fp32 (the word
    embedding is also the last gradient backward produces: the un-overlappable tail);
  * a bucket is all-reduced (RCCL, `backend="nccl"`) on the communicator's own stream as soon as its last
    expected gradient has been accumulated (`register_post_accumulate_grad_hook`), overlapping the remaining
    backward kernels;
  * `finish()` agrees on the used-parameter bitmap with one tiny MAX all-reduce, waits for the in-flight
    buckets and REBINDS every used `param.grad` to its slice of the averaged fp32 bucket — no per-parameter
    copy or scale kernels;
  * gradient accumulation (`training.update_frequency > 1`): run the first micro-batches under
    `with reducer.no_sync():` — nothing is launched or rebound, autograd keeps accumulating into `param.grad`
    — and the last one normally.
"""
import contextlib

import torch
import torch.distributed as dist

from mmf_amd import _native as _nat


def _to_f32(flat):
    """The averaged bf16 bucket as fp32 (what `param.grad` is rebound to): the HIP cast kernel on the GPU, `.float()` for the CPU host-logic tests."""
    if not flat.is_cuda:
        return flat.float()
    out = torch.empty(flat.shape, dtype=torch.float32, device=flat.device)
    _nat.cast_bf16_to_f32(flat, out)
    return out


def _wants_fp32_on_the_wire(module):
    """Parameters of embedding tables (nn.Embedding weights)."""
    ids = set()
    for m in module.modules():
        if isinstance(m, torch.nn.Embedding):
            ids.add(id(m.weight))
    return ids


class GradientReducer:
    def __init__(self, module, bucket_bytes=64 << 20, process_group=None, comm_dtype=torch.float32, fp32_params=None,
                 static_graph=True):
        """`static_graph=True` (DDP's contract of the same name): once the set of parameters that receive a gradient has been
        the same for two consecutive steps AND identical on every rank (each rank's own bitmap equals the agreed one — checked
        with the same exchange, never assumed) it is frozen — `finish()` then needs neither the bitmap exchange nor the host
        synchronisation that reading it costs, and a later deviation raises instead of hanging a collective.  A model whose
        ranks use different parameters (a head drawn per rank, as UNITERForPretraining does) simply never freezes."""
        self.module = module
        self.static_graph = static_graph
        self._stable, self._frozen, self._last_used, self._frozen_local = 0, False, None, None
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.comm_dtype = comm_dtype if comm_dtype is not None else torch.float32
        fp32_ids = _wants_fp32_on_the_wire(module) if fp32_params is None else {id(p) for p in fp32_params}
        # reverse registration order ~ order in which backward produces gradients
        params = [p for p in module.parameters() if p.requires_grad]
        params.reverse()
        self.params = params
        self.index = {id(p): i for i, p in enumerate(params)}
        self.buckets = []          # list of dict(params, offs, total, dtype)
        cur = {torch.float32: ([], 0), self.comm_dtype: ([], 0)}

        def close(dt):
            plist, _ = cur[dt]
            if plist:
                offs, total = [], 0
                for p in plist:
                    offs.append(total)
                    total += (p.numel() + 63) // 64 * 64      # slots start on 256-byte boundaries (16-byte vector loads)
                self.buckets.append(dict(params=plist, offs=offs, total=total, dtype=dt))
            cur[dt] = ([], 0)

        for p in params:
            dt = torch.float32 if (id(p) in fp32_ids or self.comm_dtype == torch.float32) else self.comm_dtype
            plist, size = cur[dt]
            plist.append(p)
            size += p.numel() * 4
            cur[dt] = (plist, size)
            if size >= bucket_bytes:
                close(dt)
        for dt in list(cur):
            close(dt)
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._bucket_of[id(p)] = bi
        self._skip = set()        # parameters without a gradient in the previous synchronised step (e.g. the pooler)
        self._sync = True
        self._handles = []
        if self.world > 1:
            for p in params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.reset()

    # ---- bookkeeping -------------------------------------------------------------------------------------------------
    def reset(self):
        self._ready = [set() for _ in self.buckets]
        self._inflight = []
        self._launched = [False] * len(self.buckets)
        self._included = set()
        self._fired = set()
        self._next = 0

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside the block neither launch nor rebind anything."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _expected(self, bi):
        return sum(1 for p in self.buckets[bi]["params"] if id(p) not in self._skip)

    def _on_grad(self, p):
        if not self._sync:
            return
        bi = self._bucket_of[id(p)]
        self._fired.add(id(p))
        if id(p) in self._skip:
            return      # was unused last step: not waited for; picked up by finish() (a straggler) if it fires this step
        self._ready[bi].add(id(p))
        self._launch_ready()

    def _launch_ready(self):
        """Collectives must be issued in the same order on every rank: buckets go out strictly in index order (which is the
        order backward fills them), each as soon as its expected gradients — an agreed set — have all arrived."""
        n = len(self.buckets)
        while self._next < n and len(self._ready[self._next]) >= self._expected(self._next):
            self._launch(self._next)
            self._next += 1

    def _pack(self, plist, offs, total, dtype, device):
        """One flat buffer in `dtype`, zero where a parameter has no gradient; filled by ONE multi-tensor (converting) copy."""
        # only gradients whose hook has fired in THIS backward: under accumulation a parameter nobody waits for may still hold
        # the partial sum of the earlier micro-batches when its bucket goes out (it is then picked up as a straggler)
        have = [(p, o) for p, o in zip(plist, offs) if p.grad is not None and id(p) in self._fired]
        dense = len(have) == len(plist) and all(p.numel() % 64 == 0 for p in plist)
        flat = (torch.empty if dense else torch.zeros)(total, dtype=dtype, device=device)
        if have:
            srcs = [p.grad for p, _ in have]
            scale = 1.0 / self.world if dtype != torch.float32 else 1.0      # the mean's 1 / world before the rounding (see module docstring)
            if flat.is_cuda and all(g.dtype == torch.float32 and g.is_contiguous() for g in srcs):
                # one HIP multi-tensor launch: every gradient read once, scaled, converted and written once (round 4 did this with
                # `_foreach_mul` + `_foreach_copy_`: a scaled fp32 copy of the whole bucket in between, ~0.7 GB of extra traffic per step)
                _nat.pack_f32_multi(srcs, [o for _, o in have], flat, scale)
            else:       # (gloo on CPU tensors: the host-logic tests)
                if scale != 1.0:
                    srcs = torch._foreach_mul(srcs, scale)
                torch._foreach_copy_([flat[o:o + p.numel()].view_as(p) for p, o in have], srcs)
        return flat, have

    def _launch(self, bi):
        self._launched[bi] = True
        b = self.buckets[bi]
        if self._expected(bi) == 0:
            return      # nobody anywhere used these parameters last step (agreed): no collective; a surprise gradient is a straggler
        device = next((p.grad.device for p in b["params"] if p.grad is not None), b["params"][0].device)
        flat, have = self._pack(b["params"], b["offs"], b["total"], b["dtype"], device)
        for p, _ in have:
            self._included.add(id(p))
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((work, flat, b["params"], b["offs"]))

    # ---- end of backward ---------------------------------------------------------------------------------------------
    def finish(self):
        """Call after the `loss.backward()` of a synchronised step: flush buckets that are still waiting (parameters
        without a gradient never report), agree on which parameters were used anywhere, wait, average, rebind."""
        if self.world <= 1 or not self._sync:
            return
        while self._next < len(self.buckets):
            self._launch(self._next)
            self._next += 1
        if self._frozen:
            local = tuple(1 if p.grad is not None else 0 for p in self.params)
            if local != self._frozen_local:
                raise RuntimeError("GradientReducer(static_graph=True): the set of parameters that receive a gradient changed "
                                   "after it had been frozen; build the reducer with static_graph=False")
            inv = 1.0 / self.world
            for work, flat, plist, offs in self._inflight:
                work.wait()
                if flat.dtype != torch.float32:
                    flat = _to_f32(flat)     # (already the mean: scaled before the rounding)
                else:
                    flat.mul_(inv)
                for off, p in zip(offs, plist):
                    p.grad = flat[off:off + p.numel()].view_as(p) if local[self.index[id(p)]] else None
            self.reset()
            return
        # stragglers: parameters that fired after their bucket went out (they were unused last step, so nobody waited)
        late = [p for p in self.params if p.grad is not None and id(p) not in self._included]
        late_ids = {id(p) for p in late}
        device = next((p.grad.device for p in self.params if p.grad is not None), self.params[0].device)
        used = torch.tensor([1 if p.grad is not None else 0 for p in self.params], dtype=torch.int32, device=device)
        late_map = torch.tensor([1 if id(p) not in self._included and p.grad is not None else 0 for p in self.params],
                                dtype=torch.int32, device=device)
        both = torch.stack([used, late_map, 1 - used])
        dist.all_reduce(both, op=dist.ReduceOp.MAX, group=self.group)
        used_any, late_any, unused_any = both[0].tolist(), both[1].tolist(), both[2].tolist()
        ranks_agree = not any(u and n for u, n in zip(used_any, unused_any))    # no parameter used on some ranks only
        late_params = [p for p, f in zip(self.params, late_any) if f]        # the same list on every rank
        late_flat = late_offs = None
        if late_params:
            late_offs, total = [], 0
            for p in late_params:
                late_offs.append(total)
                total += (p.numel() + 63) // 64 * 64
            # only the ranks on which it was late contribute it here; the others already sent theirs inside the bucket
            contrib = [p if id(p) in late_ids else None for p in late_params]
            late_flat = torch.zeros(total, dtype=torch.float32, device=device)
            src = [(p, o) for p, o in zip(contrib, late_offs) if p is not None]
            if src:
                torch._foreach_copy_([late_flat[o:o + p.numel()].view_as(p) for p, o in src], [p.grad for p, _ in src])
            dist.all_reduce(late_flat, op=dist.ReduceOp.SUM, group=self.group)
        inv = 1.0 / self.world
        rebound = set()
        for work, flat, plist, offs in self._inflight:
            work.wait()
            if flat.dtype != torch.float32:
                flat = _to_f32(flat)     # (already the mean: scaled before the rounding)
            else:
                flat.mul_(inv)           # one multiply per bucket
            for off, p in zip(offs, plist):
                if used_any[self.index[id(p)]]:
                    p.grad = flat[off:off + p.numel()].view_as(p)   # rebind, no copy: the optimizer reads the bucket slice
                    rebound.add(id(p))
                else:
                    p.grad = None
        if late_params:
            late_flat.mul_(inv)
            for off, p in zip(late_offs, late_params):
                late = late_flat[off:off + p.numel()].view_as(p)
                # A late parameter whose bucket went out holds (after the rebind above) the mean of the ranks that had it in
                # time.  If its WHOLE bucket was skipped (nobody used any of its parameters last step: no collective, nothing
                # rebound) `p.grad` is still this rank's local gradient — or None where it did not fire — and the mean is the
                # straggler sum alone: assigned, never added to the local value.
                p.grad = p.grad + late if id(p) in rebound else late
        self._skip = {id(p) for p, f in zip(self.params, used_any) if not f}
        agreed = tuple(used_any)
        self._stable = self._stable + 1 if agreed == self._last_used else 0
        self._last_used = agreed
        if self.static_graph and self._stable >= 1 and not late_params and ranks_agree:
            self._frozen = True
            self._frozen_local = agreed      # == every rank's own bitmap (ranks_agree)
        self.reset()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def parallelize_model(model, **kw):
    """TrainerDeviceMixin.parallelize_model (device.py:75-113), MI355X flavour: returns the reducer to
    call `finish()` on after backward (a no-op at world size 1)."""
    return GradientReducer(model, **kw)
