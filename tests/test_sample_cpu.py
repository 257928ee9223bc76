"""The reference's own tests of the batch object on the drop-in boundary (`Sample` / `SampleList` / `to_device`,
mmf/common/sample.py), ported from /root/reference/tests/common/test_sample.py (and the helper `build_random_sample_list` of
tests/test_utils.py:146-161), plus the constructor / assignment behaviours the models rely on."""
import random

import pytest
import torch

from mmf_amd.common.sample import Sample, SampleList, convert_batch_to_sample_list, detach_tensor, to_device


def build_random_sample_list():
    def one():
        s = Sample()
        s.x = random.randint(0, 100)
        s.y = torch.rand((5, 4))
        s.z = Sample()
        s.z.x = random.randint(0, 100)
        s.z.y = torch.rand((6, 4))
        return s
    return SampleList([one(), one()])


def test_sample_working():
    initial = Sample()
    initial.x = 1
    initial["y"] = 2
    assert initial.x == 1 and initial["x"] == 1 and initial.y == 2 and initial["y"] == 2
    initial.update({"a": 3, "b": {"c": 4}})
    assert initial.a == 3 and initial["a"] == 3
    assert initial.b.c == 4 and initial["b"].c == 4


def test_batching_of_samples():
    sl = build_random_sample_list()
    assert sl.y.shape == (2, 5, 4) and sl.z.y.shape == (2, 6, 4)
    assert isinstance(sl.x, list) and len(sl.x) == 2 and isinstance(sl.z, SampleList) and len(sl.z.x) == 2
    assert sl.get_batch_size() == 2 and sl.get_device() == torch.device("cpu")
    assert sl.fields() == ["x", "y", "z"]
    with pytest.raises(AttributeError, match="Key w not found in the SampleList"):
        sl.w
    only = sl.get_fields(["y"])
    assert only.fields() == ["y"] and torch.equal(only.y, sl.y)
    with pytest.raises(AttributeError, match="not present in SampleList"):
        sl.get_fields(["nope"])
    assert torch.equal(sl.get_field("y"), sl.y)
    nested = sl.get_item_list("z")
    assert isinstance(nested, SampleList) and nested.y.shape == (1, 2, 6, 4)


def test_samples_of_different_sizes_are_refused():
    with pytest.raises(AssertionError, match="Fields for all samples must be equally sized. a is of different sizes"):
        SampleList([Sample({"a": torch.zeros(3)}), Sample({"a": torch.zeros(4)})])
    sl = SampleList([{"a": torch.tensor(1.0)}, {"a": torch.tensor(2.0)}])      # 0-d tensors batch to [2]
    assert sl.a.shape == (2,)


def test_construction_from_pairs_and_from_a_dict():
    t = torch.arange(6).view(3, 2)
    sl = SampleList([("a", t), ("name", "vqa2")])
    assert torch.equal(sl.a, t) and sl.name == "vqa2" and sl.get_batch_size() == 3
    sl = SampleList({"a": t, "info": {"b": torch.ones(3)}})
    assert isinstance(sl.info, SampleList) and sl.info.get_batch_size() == 3 and sl.get_batch_size() == 3


def test_add_field_checks_the_batch_size_and_attribute_assignment_does_not():
    sl = SampleList({"a": torch.zeros(3, 2)})
    with pytest.raises(AssertionError, match="A tensor field to be added must have same size as existing tensor fields in SampleList"):
        sl.add_field("b", torch.zeros(4))
    sl.add_field("s", torch.tensor(1.0))          # 0-d: exempt
    sl.targets = torch.zeros(9, 5)                # plain assignment (sample.py:160-161), e.g. ViLBERT's per-pair targets
    assert sl.targets.shape == (9, 5) and sl.get_batch_size() == 3
    with pytest.raises(AssertionError):           # ... but the checked copy `to` makes refuses it, as the reference's does
        sl.to("cpu")
    empty = SampleList()
    empty.a = torch.zeros(2)
    assert empty.get_batch_size() == 2


def test_to_dict():
    sample_list = build_random_sample_list()
    sample_dict = sample_list.to_dict()
    assert isinstance(sample_dict, dict) and not isinstance(sample_dict, SampleList)
    assert not hasattr(sample_dict, "x")
    assert set(sample_dict) == {"x", "y", "z"} and set(sample_dict["z"]) == {"x", "y"} and isinstance(sample_dict["z"], dict)


def test_to_device():
    sample_list = build_random_sample_list()
    modified = to_device(sample_list, "cpu")
    assert modified.get_device() == torch.device("cpu")
    modified = to_device(sample_list, torch.device("cpu"))
    assert modified.get_device() == torch.device("cpu")
    with pytest.warns(UserWarning) if not torch.cuda.is_available() else _nullcontext():
        modified = to_device(sample_list, "cuda")
    assert modified.get_device() == (torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu"))
    double_modified = to_device(modified, modified.get_device())
    assert double_modified is modified
    custom_batch = [{"a": 1}]
    with pytest.warns(UserWarning, match="You are not returning SampleList/Sample from your dataset"):
        assert to_device(custom_batch) == custom_batch
    with pytest.raises(TypeError, match="device must be either 'str' or 'torch.device' type"):
        sample_list.to(0)


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def test_convert_batch_to_sample_list():
    batch = [{"a": torch.tensor([1.0, 1.0])}, {"a": torch.tensor([2.0, 2.0])}]
    sample_list = convert_batch_to_sample_list(batch)
    expected_a = torch.tensor([[1.0, 1.0], [2.0, 2.0]])
    assert torch.equal(expected_a, sample_list.a)
    sample_list = SampleList()
    sample_list.add_field("a", expected_a)
    parsed = convert_batch_to_sample_list([sample_list])
    assert isinstance(parsed, SampleList) and "a" in parsed and torch.equal(expected_a, parsed.a)
    batch = [{"a": [1]}, {"a": [2]}]
    sample_list = convert_batch_to_sample_list(batch)
    assert sample_list.a == [[1], [2]]
    # a SampleList filled by item assignment only (no tensor field recorded) is rebuilt with one
    raw = SampleList()
    raw["a"] = expected_a
    assert raw._get_tensor_field() is None
    assert convert_batch_to_sample_list(raw).get_batch_size() == 2


def test_detach_and_pin_memory_reach_nested_fields():
    sl = SampleList({"a": torch.zeros(2, 3, requires_grad=True) * 1.0, "n": {"b": torch.ones(2, requires_grad=True) * 2.0}, "name": "x"})
    assert sl.a.requires_grad and sl.n.b.requires_grad
    sl.detach()
    assert not sl.a.requires_grad and not sl.n.b.requires_grad and sl.name == "x"
    assert detach_tensor("text") == "text"
    moved = sl.to("cpu")
    assert moved is not sl and moved.n is not sl.n and torch.equal(moved.n.b, sl.n.b)


@pytest.mark.gpu
def test_pin_memory():
    sample_list = build_random_sample_list()
    sample_list.pin_memory()
    assert sample_list.y.is_pinned() and sample_list.z.y.is_pinned()
    assert not any(hasattr(v, "is_pinned") and v.is_pinned() for v in (sample_list.x, sample_list.z.x))


def test_batch_collator_call():
    """/root/reference/tests/common/test_batch_collator.py, ported: the collate function in front of the model."""
    from mmf_amd.common.batch_collator import BatchCollator
    batch_collator = BatchCollator("vqa2", "train")
    sample_list = build_random_sample_list()
    sample_list = batch_collator(sample_list)
    assert sample_list.dataset_name == "vqa2" and sample_list.dataset_type == "train"      # an already built sample list
    sample = Sample()
    sample.a = torch.tensor([1, 2], dtype=torch.int)
    sample_list = batch_collator([sample, sample])                                         # a list of samples
    assert torch.equal(sample_list.a, torch.tensor([[1, 2], [1, 2]], dtype=torch.int))
    sample_list = build_random_sample_list()                                               # the IterableDataset case
    new_sample_list = batch_collator([sample_list])
    assert new_sample_list == sample_list
