"""The trainer-side helpers that sit right after the path: which parameters the optimizer gets
(`get_optimizer_parameters`, mmf/utils/general.py:136-164), the unused-parameter report (`check_unused_parameters`, :167-180) and
gradient clipping (`clip_gradients`, :33-50) — which, with this package's `adam_w`, is one multi-tensor HIP reduction whose clip factor
the next fused `step()` applies (mmf_amd/modules/optimizers.py::clip_grad_norm)."""
import logging

from torch import nn

logger = logging.getLogger(__name__)


def get_optimizer_parameters(model, config):
    """Parameter groups for the optimizer: the model's own `get_optimizer_parameters(config)` when it has one (VisualBERT & co.: the BERT
    recipe of mmf/utils/modeling.py:18-46), looked up through a `DataParallel` / `DistributedDataParallel` wrapper too; else all of
    `model.parameters()`.  Always returned as a list of group dicts whose "params" are lists."""
    target = model.module if isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel)) else model
    if hasattr(target, "get_optimizer_parameters"):
        parameters = target.get_optimizer_parameters(config)
    else:
        parameters = model.parameters()
    parameters = list(parameters)
    if len(parameters) == 0:
        raise ValueError("optimizer got an empty parameter list")
    if not isinstance(parameters[0], dict):
        parameters = [{"params": parameters}]
    for group in parameters:
        group["params"] = list(group["params"])
    check_unused_parameters(parameters, model, config)
    return parameters


def check_unused_parameters(parameters, model, config):
    """Log the trainable parameters no group holds (they would silently never be updated)."""
    held = {id(p) for group in parameters for p in group["params"]}
    unused = [n for n, p in model.named_parameters() if p.requires_grad and id(p) not in held]
    if unused:
        logger.info("Model parameters not used by optimizer: {}".format(" ".join(unused)))
    return unused


def clip_gradients(model, optimizer, i_iter, writer, config, scale=1.0):
    """`training.max_grad_l2_norm` / `training.clip_norm_mode: all` (general.py:33-50).  An optimizer that provides `clip_grad_norm`
    (this package's fused AdamW: the norm is one HIP launch, the scaling rides in the update kernel) is asked first."""
    max_grad_l2_norm = config.training.max_grad_l2_norm
    clip_norm_mode = config.training.clip_norm_mode
    if max_grad_l2_norm is None:
        return None
    if clip_norm_mode != "all":
        raise NotImplementedError("Clip norm mode %s not implemented" % clip_norm_mode)
    if hasattr(optimizer, "clip_grad_norm"):
        norm = optimizer.clip_grad_norm(max_grad_l2_norm * scale)
    else:
        norm = nn.utils.clip_grad_norm_(model.parameters(), max_grad_l2_norm * scale)
    if writer is not None:
        writer.add_scalars({"grad_norm": norm}, i_iter)
    return norm
