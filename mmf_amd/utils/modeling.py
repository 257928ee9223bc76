"""Optimizer parameter groups for BERT-style models (mmf/utils/modeling.py:18-72): weight decay 0.01
everywhere except biases and LayerNorm parameters; optional LR multiplier for the non-classifier
modules when fine-tuning."""
from torch import nn


def get_bert_configured_parameters(module, lr=None, weight_decay=0.01):
    if isinstance(module, nn.Module):
        param_optimizer = list(module.named_parameters())
    else:
        param_optimizer = module
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    groups = [
        {"params": [p for n, p in param_optimizer if not any(nd in n for nd in no_decay)], "weight_decay": weight_decay},
        {"params": [p for n, p in param_optimizer if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
    ]
    if lr is not None:
        for g in groups:
            g["lr"] = lr
    return groups


def get_optimizer_parameters_for_bert(module, config):
    lr = config.optimizer.params.lr
    model_config = config.model_config.get(config.model, {})
    finetune_lr_multiplier = model_config.get("finetune_lr_multiplier", 1)
    if module.config.training_head_type == "pretraining" or finetune_lr_multiplier == 1:
        return get_bert_configured_parameters(module)
    parameters = []
    for name, submodule in module.named_children():
        if name == "classifier":
            continue
        parameters += get_bert_configured_parameters(submodule, lr * finetune_lr_multiplier)
    parameters += get_bert_configured_parameters(module.classifier)
    return parameters
