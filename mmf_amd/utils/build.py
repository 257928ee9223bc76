"""`build_model(config)` as the reference does it (mmf/utils/build.py:116-151): look the class up in
the registry, construct it with its model config, `build()`, `init_losses()`."""
from mmf_amd.common.registry import registry


def build_model(config):
    model_name = config.model if hasattr(config, "model") else config["model"]
    model_class = registry.get_model_class(model_name)
    if model_class is None:
        raise RuntimeError("No model registered for name: %s" % model_name)
    model = model_class(config)
    if hasattr(model, "build"):
        model.build()
        model.init_losses()
    return model
