"""DevicePrefetcher (input side): order, content and nesting are preserved; on the GPU the batches arrive
in HBM, optionally with bf16 features, and the model consumes them."""
import pytest
import torch

from mmf_amd.common.prefetch import DevicePrefetcher
from mmf_amd.common.sample import SampleList


def batches(n, B=2):
    for i in range(n):
        yield {"input_ids": torch.full((B, 4), i), "image_feature_0": torch.full((B, 3, 8), float(i)),
               "image_info_0": {"max_features": torch.full((B,), 3)}, "dataset_name": "vqa2"}


def test_cpu_passthrough_keeps_order_and_structure():
    out = list(DevicePrefetcher(list(batches(5)), device="cpu", depth=2))
    assert len(out) == 5
    for i, b in enumerate(out):
        assert isinstance(b, SampleList) and b["dataset_name"] == "vqa2"
        assert int(b["input_ids"][0, 0]) == i and int(b["image_info_0"]["max_features"][0]) == 3
    assert list(DevicePrefetcher([], device="cpu")) == []
    with pytest.raises(ValueError):
        DevicePrefetcher([], depth=0)


@pytest.mark.gpu
def test_gpu_batches_arrive_in_order_with_optional_bf16_features():
    out = []
    for b in DevicePrefetcher(list(batches(7)), device="cuda", depth=3, feature_dtype=torch.bfloat16):
        assert b["input_ids"].is_cuda and b["image_info_0"]["max_features"].is_cuda
        assert b["image_feature_0"].dtype == torch.bfloat16 and b["input_ids"].dtype == torch.int64
        out.append((int(b["input_ids"][0, 0]), float(b["image_feature_0"][0, 0, 0])))
    assert out == [(i, float(i)) for i in range(7)]
