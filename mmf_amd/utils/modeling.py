"""Optimizer parameter groups for BERT-style models — the rule of mmf/utils/modeling.py:18-72 (HF's fine-tuning recipe): weight decay 0.01
for everything except parameters whose NAME contains `bias`, `LayerNorm.bias` or `LayerNorm.weight`; with `finetune_lr_multiplier` != 1 on
a non-pretraining head every child module except `classifier` trains at lr x multiplier.  Group order (decayed first, then exempt; children
in registration order, the classifier last) is part of the contract: optimizer checkpoints index their state by it."""
from torch import nn

_NO_DECAY_MARKERS = ("bias", "LayerNorm.bias", "LayerNorm.weight")


def _is_exempt(parameter_name):
    return any(marker in parameter_name for marker in _NO_DECAY_MARKERS)


def get_bert_configured_parameters(module, lr=None, weight_decay=0.01):
    """Two groups for `module` (an nn.Module, or an already listed `named_parameters()`)."""
    named = list(module.named_parameters()) if isinstance(module, nn.Module) else list(module)
    decayed, exempt = [], []
    for parameter_name, parameter in named:
        (exempt if _is_exempt(parameter_name) else decayed).append(parameter)
    groups = [dict(params=decayed, weight_decay=weight_decay), dict(params=exempt, weight_decay=0.0)]
    if lr is not None:
        for group in groups:
            group["lr"] = lr
    return groups


def get_optimizer_parameters_for_bert(module, config):
    base_lr = config.optimizer.params.lr
    multiplier = config.model_config.get(config.model, {}).get("finetune_lr_multiplier", 1)
    if multiplier == 1 or module.config.training_head_type == "pretraining":
        return get_bert_configured_parameters(module)           # one learning rate for the whole model
    groups = []
    for child_name, child in module.named_children():
        if child_name != "classifier":
            groups.extend(get_bert_configured_parameters(child, base_lr * multiplier))
    groups.extend(get_bert_configured_parameters(module.classifier))      # the head keeps the base learning rate
    return groups
