"""`adam_w` for the MI355X path: the reference registers `transformers.AdamW` under that key
(mmf/modules/optimizers.py:8-17); this is the same update rule as ONE multi-tensor HIP kernel per 40
parameters (`mmf_adamw_multi`), which also refreshes the bf16 weight shadows in the same pass and can fold
gradient clipping (`clip_gradients`, mmf/utils/general.py:33-50) into the update.

    optimizer = registry.get_optimizer_class("adam_w")(model.get_optimizer_parameters(config), lr=5e-5, eps=1e-8)
"""
import os

import torch

from mmf_amd import _native as nat
from mmf_amd import functional as Fn
from mmf_amd.common.registry import registry


@registry.register_optimizer("adam_w")
class AdamW(torch.optim.Optimizer):
    """transformers.AdamW signature and semantics (`correct_bias`, decoupled decay applied after the Adam step).
    `torch_mode=True` switches to torch.optim.AdamW's rule (decay first, eps outside the bias-corrected sqrt)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True, torch_mode=False,
                 capturable=False, schedule=None):
        """`capturable=True` keeps the step count (bias correction) and the LR-schedule factor in device memory, advanced
        by a one-thread kernel at the start of every `step()`, so the whole update can be replayed from a hipGraph
        (`mmf_amd.utils.graph.GraphedTrainStep(optimizer=...)`).  `schedule=("warmup_linear", warmup_steps, total_steps)`
        evaluates MMF's `warmup_linear` scheduler on the device as well (do not also attach a host-side scheduler)."""
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameters: {} - should be in [0.0, 1.0[".format(betas))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))
        self.torch_mode = torch_mode
        self._clip = None
        self.capturable = capturable
        if schedule is not None and (not capturable or schedule[0] != "warmup_linear"):
            raise ValueError("schedule=('warmup_linear', warmup_steps, total_steps) needs capturable=True")
        self._schedule = (0, 0.0, 0.0) if schedule is None else (1, float(schedule[1]), float(schedule[2]))
        self._dev_state = None
        self.grad_scale = 1.0       # every gradient is multiplied by this inside the update (a data-parallel SUM becomes the mean)
        self.external_grads = None  # id(parameter) -> tensor to read its gradient from instead of `.grad` (fp32 or bf16, contiguous, same
                                    # element count): the data-parallel step points this at its wire buffers, so the summed bf16
                                    # gradients are consumed where the all-reduce left them (no unpack / convert pass)

    def zero_grad(self, set_to_none=True):
        """What the reference's loop calls at the head of every update (mmf/trainers/core/training_loop.py:209).  Gradients are dropped, not zeroed
        (torch's default since 2.0): the backward pass writes fresh tensors, and the plain loop costs a third of torch.optim.Optimizer.zero_grad's
        per-parameter bookkeeping on the host."""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    @torch.no_grad()
    def ensure_state(self, params=None):
        """Allocate the moments of `params` (default: every parameter) now instead of at their first step: a hipGraph capture of
        `step()` must not contain the zero-fills."""
        wanted = None if params is None else {id(p) for p in params}
        for group in self.param_groups:
            for p in group["params"]:
                if (wanted is None or id(p) in wanted) and len(self.state[p]) == 0:
                    st = self.state[p]
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        if self.capturable and self._dev_state is None:
            self._dev_state = torch.zeros(2, dtype=torch.float32, device=self.param_groups[0]["params"][0].device)

    @torch.no_grad()
    def clip_grad_norm(self, max_norm):
        """Total gradient L2 norm (one deterministic multi-tensor reduction); the clipping itself is folded into the
        next `step()`.  `clip_gradients` (general.py:39-40) calls this when the optimizer provides it."""
        grads = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not grads:
            return torch.zeros(())
        norm_sq = torch.empty(1, dtype=torch.float32, device=grads[0].device)
        nat.l2norm_sq_multi([g if g.is_contiguous() else g.contiguous() for g in grads], norm_sq)
        self._clip = (norm_sq, float(max_norm))
        return norm_sq.sqrt()[0] * self.grad_scale

    @torch.no_grad()
    def advance(self, seed=None):
        """Advance the device-side step count / schedule factor now — and, in the SAME one-thread launch, the dropout seed word of a captured step
        (`mmf_step_advance`) — instead of at the start of `step()`; the caller then passes `advance=False` to `step()`."""
        if not self.capturable:
            raise RuntimeError("advance() is for capturable=True optimizers (the counters live in device memory)")
        if self._dev_state is None:
            self._dev_state = torch.zeros(2, dtype=torch.float32, device=self.param_groups[0]["params"][0].device)
        nat.step_advance(seed, self._dev_state, *self._schedule)

    def _grad_of(self, p):
        if self.external_grads is not None:
            g = self.external_grads.get(id(p))
            if g is not None:
                return g
        return p.grad

    @torch.no_grad()
    def step(self, closure=None, only=None, advance=True):
        """`only` (ids of parameters) restricts the update to a subset — the data-parallel step updates one backward stage at a time,
        each as soon as its all-reduce has landed; `advance=False` on every such call but the first of a step keeps the device-side
        step count / schedule factor from advancing more than once."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        dev_state = None
        if self.capturable:
            if self._dev_state is None:
                dev = self.param_groups[0]["params"][0].device
                self._dev_state = torch.zeros(2, dtype=torch.float32, device=dev)
            if advance:
                nat.optim_state_advance(self._dev_state, *self._schedule)
            dev_state = self._dev_state
        # Learning rate and weight decay travel per tensor, so parameter groups that share betas / eps / correct_bias (the two BERT groups of
        # mmf/utils/modeling.py:18-46 and a finetune-LR group always do) share launches: ceil(tensors / MMF_MT_MAX) launches per step instead of that per group.
        launches = {}
        native = Fn.NATIVE and torch.cuda.is_available()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if only is not None and id(p) not in only:
                    continue
                grad = self._grad_of(p)
                if grad is None:
                    continue
                if grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                # per-parameter step count, as transformers.AdamW keeps it (state["step"]): the bias correction of a parameter
                # that first receives a gradient late starts at 1, and checkpoints exchange with the reference optimizer
                st["step"] = int(st.get("step", 0)) + 1
                g = grad if grad.is_contiguous() else grad.contiguous()
                # (with the native operator library the mirror - bf16 weight shadow / packed bias slice - is looked up inside `_adamw_step`)
                launches.setdefault((st["step"], float(b1), float(b2), float(group["eps"]), bool(group["correct_bias"])), []).append(
                    (p, g, st["exp_avg"], st["exp_avg_sq"], None if native else Fn.shadows.slot(p), group["lr"], group["weight_decay"]))
        norm_sq, max_norm = self._clip if self._clip is not None else (None, 0.0)
        for (step, b1, b2, eps, correct_bias), items in sorted(launches.items(), key=lambda kv: kv[0]):       # (normally exactly one key)
            if native:      # one operator call: the launch descriptors are filled in C++ (0.5 ms of host time per step less than through ctypes)
                torch.ops.mmf_amd._adamw_step([it[0] for it in items], [it[1] for it in items], [it[2] for it in items], [it[3] for it in items],
                                              [float(it[5]) for it in items], [float(it[6]) for it in items], b1, b2, eps, step, correct_bias,
                                              1 if self.torch_mode else 0, float(self.grad_scale), norm_sq, float(max_norm), dev_state)
            else:
                nat.adamw_multi(items, b1, b2, eps, step, correct_bias, 1 if self.torch_mode else 0, self.grad_scale, norm_sq, max_norm, dev_state)
        self._clip = None
        if only is not None:
            Fn.shadows.refresh_transposed(only=[p for g_ in self.param_groups for p in g_["params"] if id(p) in only])
        else:
            Fn.shadows.refresh_transposed()     # W^T twins of the shadows the update just rewrote (dgrad GEMM operands)
        return loss

    # ---- checkpoints ---------------------------------------------------------------------------------------------------
    # `capturable=True` keeps the live step count and schedule factor on the device (they advance inside replayed hipGraphs,
    # where this Python never runs): state_dict() reads them back so that a resumed run continues the bias correction and the
    # warm-up where it stopped, and load_state_dict() seeds them — from this optimizer's own checkpoints or from a
    # transformers.AdamW / reference checkpoint (per-parameter state["step"]).
    def _device_step(self):
        if self.capturable and self._dev_state is not None:
            return int(round(float(self._dev_state[0].item())))
        return None

    def state_dict(self):
        dstep = self._device_step()
        if dstep is not None:
            # Under hipGraph replay this Python never runs, so the per-parameter counts are stale by the number of replays; the device
            # counter is the truth for the most advanced parameter, and a parameter that started late keeps its lag behind it
            # (transformers.AdamW keeps state["step"] per parameter: bias correction of a late starter begins at 1).
            counts = [int(st["step"]) for st in self.state.values() if "step" in st]
            top = max(counts) if counts else 0
            for st in self.state.values():
                if "step" in st:
                    st["step"] = max(0, dstep - (top - int(st["step"])))
        sd = super().state_dict()
        if self.capturable and self._dev_state is not None:
            sd["mmf_amd_dev_state"] = self._dev_state.detach().cpu().clone()
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        dev = state_dict.pop("mmf_amd_dev_state", None)
        super().load_state_dict(state_dict)
        steps = [int(st["step"]) for st in self.state.values() if "step" in st]
        if self.capturable:
            device = self.param_groups[0]["params"][0].device
            new = None
            if dev is not None:
                new = dev.to(device=device, dtype=torch.float32)
            elif steps:
                # a checkpoint written by the reference optimizer: the device counter continues from its step count; the
                # schedule factor is re-derived by the next optimizer_state_advance
                new = torch.tensor([float(max(steps)), 1.0], dtype=torch.float32, device=device)
            if new is not None:
                if self._dev_state is None:
                    self._dev_state = new.clone()
                else:
                    self._dev_state.copy_(new)      # IN PLACE: a hipGraph captured earlier replays against this very buffer
