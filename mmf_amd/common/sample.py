"""`Sample` / `SampleList` with the reference's behaviour (mmf/common/sample.py:23-469): ordered
dicts with attribute access that batch per-sample fields and move to a device in one call.  This
is the object the dataloader hands the model (`input_ids`, `input_mask`, `segment_ids`,
`image_feature_0`, `image_info_0.max_features`, `targets`, `dataset_name`, `dataset_type`)."""
import collections
from collections import OrderedDict

import torch


class Sample(OrderedDict):
    """One training example (sample.py:23-66): dict with attribute access."""

    def __init__(self, init_dict=None):
        super().__init__(init_dict or {})

    def __setattr__(self, key, value):
        if isinstance(value, collections.abc.Mapping):
            value = Sample(value)
        self[key] = value

    def __setitem__(self, key, value):
        if isinstance(value, collections.abc.Mapping) and not isinstance(value, Sample):
            value = Sample(value)
        super().__setitem__(key, value)

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def fields(self):
        return list(self.keys())


class SampleList(OrderedDict):
    """Batch of samples (sample.py:69-469).  Tensors are stacked along dim 0, nested mappings are
    batched recursively, everything else is kept as a list."""

    _TENSOR_FIELD_ = "__MMF_TENSOR_FIELD__"

    def __init__(self, samples=None):
        super().__init__(self)
        if samples is None:
            samples = []
        if isinstance(samples, collections.abc.Mapping):
            for k, v in samples.items():
                self.add_field(k, v)
            return
        if len(samples) == 0:
            return
        for field in samples[0].keys():
            vals = [s[field] for s in samples]
            first = vals[0]
            if isinstance(first, torch.Tensor):
                self[field] = torch.stack(vals, dim=0)
                if self._get_tensor_field() is None:
                    self._set_tensor_field(field)
            elif isinstance(first, collections.abc.Mapping):
                self[field] = SampleList(vals)
            else:
                self[field] = vals

    # attribute access ------------------------------------------------------------------------
    def __setattr__(self, key, value):
        if key.startswith("_OrderedDict"):
            super().__setattr__(key, value)
        else:
            self.add_field(key, value)

    def __getattr__(self, key):
        if key.startswith("_OrderedDict") or key.startswith("__"):
            raise AttributeError(key)
        if key not in self:
            raise AttributeError("Key {} not found in the SampleList. Valid choices are {}".format(key, self.fields()))
        return self[key]

    def _get_tensor_field(self):
        return self.__dict__.get(SampleList._TENSOR_FIELD_, None)

    def _set_tensor_field(self, value):
        self.__dict__[SampleList._TENSOR_FIELD_] = value

    # reference API -----------------------------------------------------------------------------
    def fields(self):
        return list(self.keys())

    def get_batch_size(self):
        tf = self._get_tensor_field()
        assert tf is not None, "There is no tensor yet in SampleList"
        return self[tf].size(0)

    def add_field(self, field, data):
        if isinstance(data, collections.abc.Mapping) and not isinstance(data, SampleList):
            data = SampleList(data)
        if isinstance(data, torch.Tensor):
            tf = self._get_tensor_field()
            if tf is None:
                self._set_tensor_field(field)
            elif field != tf and data.dim() > 0 and data.size(0) != self[tf].size(0):
                raise AssertionError(
                    "A tensor field to be added must have same size as existing tensor fields in SampleList. "
                    "Passed size: {}, Required size: {}".format(data.size(0), self[tf].size(0)))
        self[field] = data

    def get_field(self, field):
        return self[field]

    def get_fields(self, fields):
        out = SampleList()
        for f in fields:
            if f not in self:
                raise AttributeError("{} not present in SampleList. Valid choices are {}".format(f, self.fields()))
            out.add_field(f, self[f])
        return out

    def copy(self):
        out = SampleList()
        for f in self.fields():
            out.add_field(f, self[f])
        return out

    def to(self, device, non_blocking=True):
        """Move every tensor field (recursively) to `device` (sample.py:326-356)."""
        if not isinstance(device, torch.device):
            if not isinstance(device, str):
                raise TypeError("device must be either 'str' or 'torch.device' type, {} found".format(type(device)))
            device = torch.device(device)
        out = self.copy()
        for f in out.fields():
            v = out[f]
            if hasattr(v, "to"):
                out[f] = v.to(device, non_blocking=non_blocking) if isinstance(v, torch.Tensor) else v.to(device)
        return out

    def pin_memory(self):
        for f in self.fields():
            if hasattr(self[f], "pin_memory"):
                self[f] = self[f].pin_memory()
        return self

    def detach(self):
        for f in self.fields():
            if isinstance(self[f], torch.Tensor):
                self[f] = self[f].detach()
        return self

    def to_dict(self):
        out = {}
        for f in self.fields():
            out[f] = self[f].to_dict() if hasattr(self[f], "to_dict") else self[f]
        return out


def to_device(sample_list, device="cuda"):
    """mmf/common/sample.py:425-469."""
    if isinstance(sample_list, collections.abc.Mapping) and not isinstance(sample_list, SampleList):
        sample_list = SampleList(sample_list)
    if not isinstance(sample_list, SampleList):
        return sample_list
    return sample_list.to(device)
