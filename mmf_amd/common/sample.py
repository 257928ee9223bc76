"""`Sample` / `SampleList` with the reference's behaviour (mmf/common/sample.py:23-469): ordered
dicts with attribute access that batch per-sample fields and move to a device in one call.  This
is the object the dataloader hands the model (`input_ids`, `input_mask`, `segment_ids`,
`image_feature_0`, `image_info_0.max_features`, `targets`, `dataset_name`, `dataset_type`)."""
import collections
import warnings
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
        if isinstance(samples, collections.abc.Mapping):        # a dict of already batched fields (sample.py:144-150)
            for k, v in samples.items():
                self.add_field(k, v)
            return
        if len(samples) == 0:
            return
        if isinstance(samples[0], (tuple, list)) and len(samples[0]) > 0 and isinstance(samples[0][0], str):
            for pair in samples:                                # (key, batched value) pairs (sample.py:136-142)
                self.add_field(pair[0], pair[1])
            return
        for field in samples[0].keys():
            vals = [s[field] for s in samples]
            first = vals[0]
            if isinstance(first, torch.Tensor):
                for v in vals:                                  # sample.py:118-128: same leading size in every sample (0-d tensors exempt)
                    if isinstance(v, torch.Tensor) and v.dim() != 0 and v.size(0) != first.size(0):
                        raise AssertionError("Fields for all samples must be equally sized. {} is of different sizes".format(field))
                self[field] = torch.stack(vals, dim=0)
                if self._get_tensor_field() is None:
                    self._set_tensor_field(field)
            elif isinstance(first, collections.abc.Mapping):
                self[field] = SampleList(vals)
            else:
                self[field] = vals

    # attribute access ------------------------------------------------------------------------
    def __setattr__(self, key, value):
        """sample.py:160-161: a plain item assignment — unlike `add_field` it does not compare batch sizes (so a model may attach, say,
        per-pair targets).  The first tensor assigned this way also becomes the batch-size / device field, as with `add_field`."""
        if key.startswith("_OrderedDict"):
            super().__setattr__(key, value)
            return
        if isinstance(value, collections.abc.Mapping) and not isinstance(value, SampleList):
            value = SampleList(value)
        self[key] = value
        if isinstance(value, torch.Tensor) and self._get_tensor_field() is None:
            self._set_tensor_field(key)

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

    def get_device(self):
        """Device of the batch = device of its first tensor field (sample.py:182-187)."""
        tf = self._get_tensor_field()
        assert tf is not None, "No tensor field in sample list, available keys: {}".format(self.fields())
        return self[tf].device

    def get_item_list(self, key):
        """A SampleList holding only the (nested) field `key` (sample.py:189-203)."""
        return SampleList([self[key]])

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
            self[f] = detach_tensor(self[f])         # tensors and nested sample lists alike (sample.py:372-378)
        return self

    def to_dict(self):
        out = {}
        for f in self.fields():
            out[f] = self[f].to_dict() if hasattr(self[f], "to_dict") else self[f]
        return out


def convert_batch_to_sample_list(batch):
    """sample.py:400-420: what a collated batch becomes before the model sees it — a one-element list holding a SampleList is unwrapped,
    anything else that is not a SampleList is batched, and a SampleList built without `add_field` (no tensor field recorded) is rebuilt."""
    sample_list = batch
    if isinstance(batch, list) and len(batch) == 1 and isinstance(batch[0], SampleList):
        sample_list = batch[0]
    elif not isinstance(batch, SampleList):
        sample_list = SampleList(batch)
    if sample_list._get_tensor_field() is None:
        sample_list = SampleList(sample_list.to_dict())
    return sample_list


def to_device(sample_list, device="cuda"):
    """sample.py:425-455.  Anything that is not a mapping / SampleList is handed back with a warning (the caller moves its own tensors); `cuda`
    without a GPU falls back to the host with a warning, as in the reference — the MODEL then refuses host tensors (there is no CPU arithmetic
    path here); a batch that already sits on `device` is returned as the same object."""
    if isinstance(sample_list, collections.abc.Mapping):
        sample_list = convert_batch_to_sample_list(sample_list)
    if not isinstance(sample_list, SampleList):
        warnings.warn("You are not returning SampleList/Sample from your dataset. MMF expects you to move your tensors to cuda yourself.")
        return sample_list
    if isinstance(device, str):
        device = torch.device(device)
    if device.type == "cuda" and not torch.cuda.is_available():
        warnings.warn("Selected device is cuda, but it is NOT available!!! Falling back on cpu.")
        device = torch.device("cpu")
    if sample_list.get_device() != device:
        sample_list = sample_list.to(device)
    return sample_list


def detach_tensor(tensor):
    """sample.py:457-469: `.detach()` of a tensor, SampleList or Report; anything else unchanged."""
    if hasattr(tensor, "detach"):
        tensor = tensor.detach()
    return tensor
