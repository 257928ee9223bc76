"""`MMFLoss` / `Losses` key and error behaviour, as the reference's own tests pin it (tests/modules/test_losses.py:57-103): empty
parameters raise, a string or a {type} mapping names the loss, results come back keyed "{dataset_type}/{dataset_name}/{loss}" and a
loss that returns a dict fans out into "{...}/{loss}/{child}"."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

import mmf_amd  # noqa: F401
from mmf_amd.common.registry import registry
from mmf_amd.common.sample import SampleList
from mmf_amd.modules import losses


@registry.register_loss("mse_mae")
class MSEAndMAELoss(nn.Module):
    def forward(self, sample_list, model_output):
        targets, scores = sample_list["targets"], model_output["scores"]
        return {"mse": F.mse_loss(scores, targets), "mae": F.l1_loss(scores, targets)}


@registry.register_loss("constant_one")
class ConstantOne(nn.Module):
    def forward(self, sample_list, model_output):
        return torch.tensor(1.0)


def test_mmf_loss():
    with pytest.raises(ValueError):
        losses.MMFLoss()
    with pytest.raises(ValueError):
        losses.MMFLoss({})
    assert losses.MMFLoss({"type": "constant_one"}).name == "constant_one"
    assert losses.MMFLoss("constant_one").name == "constant_one"
    with pytest.raises(AssertionError):
        losses.MMFLoss([])
    with pytest.raises(ValueError):
        losses.MMFLoss("not_a_registered_loss")
    sl = SampleList(dict(dataset_type="val", dataset_name="vqa2"))
    out = losses.MMFLoss("constant_one")(sl, {})
    out_from_dict = losses.MMFLoss({"type": "constant_one"})(sl, {})
    assert list(out) == ["val/vqa2/constant_one"] and torch.equal(out["val/vqa2/constant_one"], torch.tensor([1.0]))
    assert out.keys() == out_from_dict.keys()


def test_mmf_dict_loss():
    torch.manual_seed(1234)
    t = torch.rand((1, 768))
    sl = SampleList(dict(dataset_type="val", dataset_name="vqa2", targets=t))
    out = losses.MMFLoss("mse_mae")(sl, {"scores": t})
    assert out["val/vqa2/mse_mae/mse"].item() == 0.0 and out["val/vqa2/mse_mae/mae"].item() == 0.0


def test_losses_skips_without_targets_and_registers_the_result():
    ls = losses.Losses([{"type": "mse_mae"}])
    with pytest.warns(UserWarning, match="targets"):
        assert ls(SampleList(dict(dataset_type="val", dataset_name="vqa2")), {}) == {}
    t = torch.ones(2, 3)
    sl = SampleList(dict(dataset_type="train", dataset_name="vqa2", targets=t))
    out = ls(sl, {"scores": t + 1.0})
    assert sorted(out) == ["train/vqa2/mse_mae/mae", "train/vqa2/mse_mae/mse"] and out["train/vqa2/mse_mae/mse"].item() == 1.0
    assert registry.get("losses.vqa2.train") is out
