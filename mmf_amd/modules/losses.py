"""`Losses` / `MMFLoss` / `LogitBinaryCrossEntropy` with the reference's behaviour
(mmf/modules/losses.py:52-251): losses come back as a dict keyed
"{dataset_type}/{dataset_name}/{loss_name}".  `logit_bce` runs on the HIP kernels."""
import collections
import warnings

import torch
from torch import nn

from mmf_amd import fp32_train as F32T
from mmf_amd import functional as Fn
from mmf_amd import ops  # noqa: F401  (registers torch.ops.mmf_amd.*)
from mmf_amd.common.registry import registry


class Losses(nn.Module):
    """losses.py:52-131."""

    def __init__(self, loss_list):
        super().__init__()
        self.losses = nn.ModuleList()
        config = registry.get("config")
        self._evaluation_predict = False
        if config:
            self._evaluation_predict = config.get("evaluation", {}).get("predict", False)
        for loss in loss_list:
            self.losses.append(MMFLoss(loss))

    @torch.jit.unused      # losses are attached by BaseModel.__call__ (base_model.py:305-337), outside the scripted forward
    def forward(self, sample_list, model_output):
        output = {}
        if "targets" not in sample_list:
            if not self._evaluation_predict:
                warnings.warn("Sample list has not field 'targets', are you sure that your ImDB has labels? you may have "
                              "wanted to run with evaluation.predict=true")
            return output
        for loss in self.losses:
            output.update(loss(sample_list, model_output))
        registry.register("losses.%s.%s" % (sample_list["dataset_name"], sample_list["dataset_type"]), output)
        return output


class MMFLoss(nn.Module):
    """losses.py:132-222."""

    def __init__(self, params=None):
        super().__init__()
        if params is None:
            params = {}
        is_mapping = isinstance(params, collections.abc.Mapping)
        if is_mapping:
            if "type" not in params:
                raise ValueError("Parameters to loss must have 'type' field to specify type of loss to instantiate")
            loss_name = params["type"]
        else:
            assert isinstance(params, str), "loss must be a string or dictionary with 'type' key"
            loss_name = params
        self.name = loss_name
        loss_class = registry.get_loss_class(loss_name)
        if loss_class is None:
            raise ValueError("No loss named %s is registered to registry" % loss_name)
        loss_params = params.get("params", {}) if is_mapping else {}
        self.loss_criterion = loss_class(**loss_params)

    @torch.jit.unused      # losses are attached by BaseModel.__call__ (base_model.py:305-337), outside the scripted forward
    def forward(self, sample_list, model_output):
        loss_dict = {}
        datasets = getattr(self.loss_criterion, "datasets", None)
        if isinstance(datasets, list) and sample_list["dataset_name"] not in datasets:
            return loss_dict
        loss_result = self.loss_criterion(sample_list, model_output)
        if not isinstance(loss_result, collections.abc.Mapping):
            loss_result = {"": loss_result}
        for child_name, child in loss_result.items():
            if not isinstance(child, torch.Tensor):
                child = torch.tensor(child, dtype=torch.float)
            if child.dim() == 0:
                child = child.view(1)
            key = "%s/%s/%s" % (sample_list["dataset_type"], sample_list["dataset_name"], self.name)
            key = "%s/%s" % (key, child_name) if child_name else key
            loss_dict[key] = child
        return loss_dict


@registry.register_loss("logit_bce")
class LogitBinaryCrossEntropy(nn.Module):
    """mean(BCEWithLogits(scores, targets)) * targets.size(1)   (losses.py:225-251)."""

    @torch.jit.unused      # losses are attached by BaseModel.__call__ (base_model.py:305-337), outside the scripted forward
    def forward(self, sample_list, model_output):
        if F32T.active():       # mmf_amd.fp32_training(): the fp32 gradient (the operator's own backward hands bf16-rounded values on)
            return F32T.logit_bce(model_output["scores"], sample_list["targets"])
        return torch.ops.mmf_amd.logit_bce(model_output["scores"], sample_list["targets"])


@registry.register_loss("cross_entropy")
class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss(**params)(scores, targets)   (losses.py:595-602).  HIP kernels; supports
    `ignore_index` and mean reduction (the reference's defaults)."""

    def __init__(self, **params):
        super().__init__()
        extra = set(params) - {"ignore_index", "reduction"}
        if extra or params.get("reduction", "mean") != "mean":
            raise NotImplementedError("cross_entropy: only ignore_index / reduction='mean' are built (got %s)" % sorted(params))
        self.ignore_index = int(params.get("ignore_index", -100))

    @torch.jit.unused      # losses are attached by BaseModel.__call__ (base_model.py:305-337), outside the scripted forward
    def forward(self, sample_list, model_output):
        return Fn.CrossEntropyFn.apply(model_output["scores"], sample_list["targets"], self.ignore_index)


@registry.register_loss("m4c_decoding_bce_with_mask")
class M4CDecodingBCEWithMaskLoss(nn.Module):
    """losses.py:575-592: BCE over the decoding steps, weighted by `train_loss_mask`, normalised by max(sum(mask), 1)."""

    @torch.jit.unused      # losses are attached by BaseModel.__call__ (base_model.py:305-337), outside the scripted forward
    def forward(self, sample_list, model_output):
        scores = model_output["scores"]
        targets = sample_list["targets"]
        loss_mask = sample_list["train_loss_mask"]
        assert scores.dim() == 3 and loss_mask.dim() == 2
        return Fn.DecodingBCEWithMaskFn.apply(scores, targets, loss_mask)
