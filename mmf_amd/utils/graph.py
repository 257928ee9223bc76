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

import torch

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


class GraphedTrainStep:
    """`optimizer` (an `adam_w` built with `capturable=True`) puts the parameter update into the graph as well: one
    replay = one full training step.  The optimizer then also keeps the bf16 weight shadows current, so no cast kernel
    is captured; without it the casts are captured so that replays see updates made between them."""

    def __init__(self, model, batch, warmup=3, loss_of=None, optimizer=None):
        self.model = model
        self.optimizer = optimizer
        if optimizer is not None and not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs an optimizer whose step reads its counters from device memory (capturable=True)")
        self.loss_of = loss_of or (lambda out: sum(v.sum() for v in out["losses"].values()))
        release_autograd_state()
        self.static_batch = _clone_batch(batch)
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = next(model.parameters()).device
        self.seed = torch.zeros(1, dtype=torch.int32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), Fn.dropout_keys.graph_mode(self.seed):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        model.zero_grad(set_to_none=True)
        if optimizer is None:
            Fn.shadows.clear()   # so the weight-shadow casts are part of the captured step
        with Fn.dropout_keys.graph_mode(self.seed):
            with torch.cuda.graph(self.graph):
                self.out, self.loss = self._eager()

    def _eager(self):
        Fn.nat.seed_advance(self.seed)
        out = self.model(self.static_batch)
        loss = self.loss_of(out)
        # torch.autograd.grad instead of loss.backward(): AccumulateGrad nodes are bound to the stream they were first
        # created on; if an earlier eager step created them on the legacy default stream, running them inside the
        # capture drags that stream into it and hipStreamEndCapture crashes.  Capturing the gradients directly keeps
        # every captured node on the capture stream.
        grads = torch.autograd.grad(loss, self.params, allow_unused=True)
        for p, g in zip(self.params, grads):
            p.grad = g
        if self.optimizer is not None:
            self.optimizer.step()
        return out, loss

    def __call__(self, batch=None):
        if batch is not None:
            _copy_batch(self.static_batch, batch)
        self.graph.replay()
        return self.loss
