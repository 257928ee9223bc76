"""The DataLoader's `collate_fn` of the reference (mmf/common/batch_collator.py:5-15): per-sample `Sample`s (or an already batched
SampleList from an iterable dataset) become the SampleList the model reads, stamped with the `dataset_name` / `dataset_type` that
`VisualBERT.forward` and the losses key their outputs on (visual_bert.py:567-590, losses.py:212-214)."""
from mmf_amd.common import sample as _sample


class BatchCollator:
    """`BatchCollator(dataset_name, dataset_type)(batch) -> SampleList`."""

    __slots__ = ("_stamps",)

    def __init__(self, dataset_name, dataset_type):
        self._stamps = (("dataset_name", dataset_name), ("dataset_type", dataset_type))

    def __call__(self, batch):
        collated = _sample.convert_batch_to_sample_list(batch)
        for field, value in self._stamps:        # plain attribute assignment, as the reference stamps them (no batch-size check on strings)
            setattr(collated, field, value)
        return collated
