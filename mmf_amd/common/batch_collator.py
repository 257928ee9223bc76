"""The DataLoader's `collate_fn` of the reference (mmf/common/batch_collator.py:5-15): per-sample `Sample`s (or an already batched
SampleList from an iterable dataset) become the SampleList the model reads, stamped with the `dataset_name` / `dataset_type` that
`VisualBERT.forward` and the losses key their outputs on (visual_bert.py:567-590, losses.py:212-214)."""
from mmf_amd.common.sample import convert_batch_to_sample_list


class BatchCollator:
    def __init__(self, dataset_name, dataset_type):
        self._dataset_name = dataset_name
        self._dataset_type = dataset_type

    def __call__(self, batch):
        sample_list = convert_batch_to_sample_list(batch)
        sample_list.dataset_name = self._dataset_name
        sample_list.dataset_type = self._dataset_type
        return sample_list
