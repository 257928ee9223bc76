"""Host-side helpers of the graphed training steps (mmf_amd/utils/graph.py) that need no GPU."""
import torch

from mmf_amd.utils.graph import total_loss


def test_total_loss_is_the_trainers_sum_and_adds_no_kernel_for_a_single_scalar():
    """mmf/trainers/core/training_loop.py:199-213 differentiates the sum over the loss dict of every entry's mean; the losses of this package are
    scalars (MMFLoss hands them on with shape [1], like the reference).  One entry: the tensor itself as a 0-dim VIEW (no reduction, no add)."""
    w = torch.tensor([2.0], requires_grad=True)
    one = {"losses": {"train/vqa2/logit_bce": w * 3.0}}
    t = total_loss(one)
    assert t.dim() == 0 and float(t) == 6.0 and t._base is not None          # a view of the [1]-shaped loss
    t.backward()
    assert float(w.grad) == 3.0
    scalar = {"losses": {"a": torch.tensor(1.5)}}
    assert total_loss(scalar).dim() == 0 and float(total_loss(scalar)) == 1.5
    many = {"losses": {"a": torch.tensor([1.0]), "b": torch.tensor(2.5), "c": torch.tensor([[0.5, 0.5]])}}
    assert float(total_loss(many)) == 4.5
