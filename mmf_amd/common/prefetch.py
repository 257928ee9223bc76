"""Input side of the hot path: the reference moves every batch to the GPU synchronously inside
`BaseModel.__call__` -> `to_device` (mmf/models/base_model.py:305-337, mmf/common/sample.py:326-356, 425-469): 26.6 MB of
fp32 region features per step over PCIe before the first kernel can start.  `DevicePrefetcher` keeps `depth` batches
in flight instead: each batch is staged in pinned host memory and copied on a dedicated HIP stream while the previous
step computes; `__next__` only makes the compute stream wait on that copy's event.

    for batch in DevicePrefetcher(loader, device="cuda", depth=2):      # batch: SampleList already in HBM
        step(batch)

Only tensors move; everything else in the SampleList (dataset_name, ...) is carried over.  `feature_dtype=torch.bfloat16`
converts the floating-point feature fields on the device right after the copy, halving what the embedding GEMM reads
(the GEMM takes fp32 or bf16 features).

`trim_text_padding` (also `DevicePrefetcher(trim_text_padding=8)`): real VQA batches are ~90 % text padding (questions of 8 - 24 tokens in 128
positions).  The reference masks padded keys and never compacts (mmf/models/visual_bert.py:94-106): every padded position is embedded, projected,
attended FROM and normalised in all 12 layers, and thrown away.  Text columns that NO sample of the batch uses can be cut on the host before the copy
without changing any result: a masked key's probability is exp(-10000 + ...) = 0 exactly (fp32 and bf16 alike), a padded query's output feeds
nothing (the `vqa` pooling reads position `input_mask.sum(1) - 2`, ViLBERT / MMBT read position 0, masked-LM labels are -1 there), so scores, losses
and every parameter gradient are those of the untrimmed batch (tests/test_text_padding_*.py: oracle on both batches, HIP path on both batches).
Position ids of the text are `arange(T)` and the regions' do not depend on T (mmf/modules/embeddings.py:423-459), so cutting the TAIL keeps them."""
import collections

import torch

from mmf_amd.common.sample import SampleList


def _map_tensors(obj, fn):
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, SampleList):
        out = SampleList()
        for k in obj.fields():
            out[k] = _map_tensors(obj[k], fn)
        return out
    if isinstance(obj, collections.abc.Mapping):
        return SampleList({k: _map_tensors(v, fn) for k, v in obj.items()})
    return obj


TEXT_FIELDS = ("input_ids", "input_mask", "segment_ids", "lm_label_ids")


def used_text_length(mask, multiple=8):
    """Number of leading text columns to keep: past the last column ANY sample of the batch uses, rounded up to `multiple` (a handful of captured
    step shapes instead of one per length), never more than the batch has.  Reads the mask on the host: free for a batch the DataLoader just
    collated, a device-to-host synchronisation for a batch already in HBM (trim before the copy: `DevicePrefetcher(trim_text_padding=...)`)."""
    if multiple < 1:
        raise ValueError("multiple must be >= 1")
    T = int(mask.shape[-1])
    used = mask.reshape(-1, T).ne(0).any(0)
    cols = used.nonzero()
    last = int(cols.max()) + 1 if cols.numel() else 1
    return min(T, -(-last // multiple) * multiple)


def trim_text_padding(batch, multiple=8, mask_field="input_mask", text_fields=TEXT_FIELDS):
    """A SampleList whose text fields (`[..., T]` tensors named in `text_fields`) keep only the columns some sample uses (see the module docstring);
    everything else is carried over untouched.  Returns `batch` itself when nothing can be cut."""
    mask = batch[mask_field]
    T = int(mask.shape[-1])
    keep = used_text_length(mask, multiple)
    if keep >= T:
        return batch
    out = SampleList()
    for k in (batch.fields() if isinstance(batch, SampleList) else batch.keys()):
        v = batch[k]
        if k in text_fields and isinstance(v, torch.Tensor) and v.dim() >= 2 and v.shape[-1] == T:
            v = v[..., :keep].contiguous()
        out.add_field(k, v)
    return out


class DevicePrefetcher:
    def __init__(self, loader, device="cuda", depth=2, feature_fields=("image_feature_0",), feature_dtype=None, trim_text_padding=0):
        """`trim_text_padding=m` (> 0): cut the text columns no sample uses, rounded up to a multiple of m, on the HOST batch before it is pinned and
        copied (less PCIe traffic as well); the consumer then sees batches of a few different text lengths (`BucketedTrainStep`)."""
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.trim = int(trim_text_padding or 0)
        self.loader = loader
        self.device = torch.device(device)
        self.depth = depth
        self.feature_fields = tuple(feature_fields)
        self.feature_dtype = feature_dtype
        self._cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self._cuda else None

    def _stage(self, batch):
        """Pin (if needed) and start the asynchronous copy of one batch on the copy stream."""
        if not isinstance(batch, SampleList):
            batch = SampleList(batch)
        if self.trim > 0 and "input_mask" in batch and not batch["input_mask"].is_cuda:
            batch = trim_text_padding(batch, self.trim)
        if not self._cuda:
            return batch, None
        host = _map_tensors(batch, lambda t: t if (t.is_cuda or t.is_pinned()) else t.pin_memory())
        with torch.cuda.stream(self.stream):
            dev = _map_tensors(host, lambda t: t.to(self.device, non_blocking=True))
            if self.feature_dtype is not None:
                for f in self.feature_fields:
                    if f in dev and isinstance(dev[f], torch.Tensor) and dev[f].is_floating_point():
                        dev[f] = dev[f].to(self.feature_dtype)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return (dev, host), ev      # the pinned source must outlive the copy

    def __iter__(self):
        it = iter(self.loader)
        queue = collections.deque()
        try:
            while len(queue) < self.depth:
                queue.append(self._stage(next(it)))
        except StopIteration:
            pass
        while queue:
            staged, ev = queue.popleft()
            try:
                queue.append(self._stage(next(it)))
            except StopIteration:
                pass
            if ev is None:
                yield staged
                continue
            dev, _host = staged
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            _map_tensors(dev, lambda t: (t.record_stream(cur), t)[1])   # the caching allocator must not recycle it early
            yield dev

    def __len__(self):
        return len(self.loader)
