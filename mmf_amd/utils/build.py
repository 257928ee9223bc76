"""`build_model(config)` as the reference does it (mmf/utils/build.py:116-151): look the class up in
the registry, construct it with its model config, `build()`, `init_losses()`; `build_optimizer` / `build_scheduler` (:405-452, :469-485):
the optimizer (a `torch.optim` class by name, else a registered one such as this package's fused `adam_w`) over the model's parameter
groups, and the registered LR scheduler."""
import warnings

import torch

from mmf_amd.common.registry import registry


def build_model(config):
    model_name = config.model if hasattr(config, "model") else config["model"]
    model_class = registry.get_model_class(model_name)
    if model_class is None:
        raise RuntimeError("No model registered for name: %s" % model_name)
    model = model_class(config)
    if hasattr(model, "build"):
        # build.py:132-150 lets the main rank download / build first and the others wait at a barrier (collective C3 of SURVEY.md
        # section 2.3): building here touches no shared files and downloads nothing, so every rank builds at once, without a barrier
        if hasattr(model_class, "load_requirements"):
            model_class.load_requirements(model_class, config=config)
        model.build()
        model.init_losses()
    return model


def build_optimizer(model, config):
    optimizer_config = config.optimizer
    if "type" not in optimizer_config:
        raise ValueError("Optimizer attributes must have a 'type' key specifying the type of optimizer. (Custom or PyTorch, e.g. 'adam_w' or 'SGD')")
    optimizer_type = optimizer_config.type
    if "params" not in optimizer_config:
        warnings.warn("optimizer attributes has no params defined, defaulting to {}.")
    params = optimizer_config.get("params", {})
    if hasattr(torch.optim, optimizer_type):
        optimizer_class = getattr(torch.optim, optimizer_type)
    else:
        optimizer_class = registry.get_optimizer_class(optimizer_type)
        if optimizer_class is None:
            raise ValueError("No optimizer class of type {} present in either torch or registered to registry".format(optimizer_type))
    if optimizer_config.get("enable_state_sharding", False):
        raise NotImplementedError("optimizer.enable_state_sharding (fairscale OSS) is outside this package's path: the replica per GPU keeps its "
                                  "whole optimizer state (1.4 GB of 288 GB)")
    from mmf_amd.utils.general import get_optimizer_parameters
    parameters = get_optimizer_parameters(model, config)
    return optimizer_class(parameters, **params)


def build_scheduler(optimizer, config):
    scheduler_config = config.get("scheduler", {})
    if "type" not in scheduler_config:
        warnings.warn("No type for scheduler specified even though lr_scheduler is True, setting default to 'Pythia'")
    scheduler_type = scheduler_config.get("type", "pythia")
    if "params" not in scheduler_config:
        warnings.warn("scheduler attributes has no params defined, defaulting to {}.")
    params = scheduler_config.get("params", {})
    scheduler_class = registry.get_scheduler_class(scheduler_type)
    if scheduler_class is None:
        raise ValueError("No scheduler class of type {} registered (this package registers 'warmup_linear')".format(scheduler_type))
    return scheduler_class(optimizer, **params)
