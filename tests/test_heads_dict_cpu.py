"""The reference's tests/models/transformers/test_heads_dict.py ported (host logic of `build_heads_dict` / `HeadsDict`,
mmf/models/transformers/heads/utils.py): the heads run as dry runs on the kernel-wrapper stubs of tests/native_stub.py — construction from
list and mapping configs, which heads a task runs, how their losses are keyed and merged."""
import pytest
import torch

from mmf_amd.common.sample import Sample
from mmf_amd.models.transformers.heads.utils import HeadsDict, build_heads_dict, compute_masked_hidden
from mmf_amd.modules.losses import MMFLoss
from tests import native_stub


@pytest.fixture
def inputs():
    sample_list = Sample()
    sample_list["targets"] = torch.rand((1, 2))
    sample_list["dataset_type"] = "test"
    sample_list["dataset_name"] = "test_dataset"
    sample_list["is_correct"] = torch.ones((1,), dtype=torch.long)
    return sample_list, torch.rand(size=(1, 1, 768)), {"test_cls": MMFLoss("logit_bce")}


def _run(heads_dict, task, inputs):
    sample_list, model_output, _ = inputs
    assert isinstance(heads_dict, HeadsDict)
    with native_stub.installed():
        out = heads_dict.forward(task, model_output, sample_list)
    assert isinstance(out, dict) and "losses" in out
    return out


def test_constructor_on_dict_confs(inputs):
    heads_dict = build_heads_dict({"test": {"type": "mlp", "loss": "test_cls"}}, ["test"], inputs[2])
    assert "test/test_dataset/logit_bce" in _run(heads_dict, "test", inputs)["losses"]


def test_constructor_on_list_confs(inputs):
    heads_dict = build_heads_dict([{"type": "mlp", "loss": "test_cls"}], [], inputs[2])
    assert "test/test_dataset/logit_bce" in _run(heads_dict, None, inputs)["losses"]


def test_constructor_on_multiple_losses_per_task(inputs):
    heads_dict = build_heads_dict({"test": [{"type": "mlp", "loss": "test_cls"}, {"type": "itm"}]}, ["test"], inputs[2])
    losses = _run(heads_dict, "test", inputs)["losses"]
    assert "test/test_dataset/logit_bce" in losses and "itm_loss" in losses


def test_constructor_on_multiple_tasks(inputs):
    conf = {"test": {"type": "mlp", "loss": "test_cls"}, "other_task": {"type": "itm"}, "third_task": {"type": "mlm"}}
    heads_dict = build_heads_dict(conf, ["test", "other_task"], inputs[2])
    assert sorted(heads_dict.heads.keys()) == ["other_task", "test"]          # only the tasks asked for are built
    losses = _run(heads_dict, "other_task", inputs)["losses"]
    assert "test/test_dataset/logit_bce" not in losses and "itm_loss" in losses


def test_constructor_on_multiple_loss_list(inputs):
    heads_dict = build_heads_dict([{"type": "mlp", "loss": "test_cls"}, {"type": "itm"}], [], inputs[2])
    losses = _run(heads_dict, None, inputs)["losses"]
    assert "test/test_dataset/logit_bce" in losses and "itm_loss" in losses


def test_errors_and_the_masked_row_selection(inputs):
    with pytest.raises(ValueError, match="No head defined for missing"):
        build_heads_dict({"test": {"type": "mlp", "loss": "test_cls"}}, ["missing"], inputs[2])
    nameless = build_heads_dict([{"type": "mlp"}], [], inputs[2])             # scores without a loss to apply
    with native_stub.installed(), pytest.raises(ValueError, match="must either define a 'loss'"):
        nameless.forward(None, inputs[1], inputs[0])
    hidden = torch.arange(2 * 3 * 4, dtype=torch.float32).view(2, 3, 4)
    mask = torch.tensor([[True, False, True], [False, False, True]])
    rows = compute_masked_hidden(hidden, mask)
    assert rows.shape == (3, 4) and torch.equal(rows, torch.stack([hidden[0, 0], hidden[0, 2], hidden[1, 2]]))
