"""/root/reference/tests/models/test_visual_bert.py:71-98 (`TestVisualBertPretraining.test_pretrained_model`), its construction of the batch
kept: a SampleList filled with `add_field`, no `image_info_0` at all, `lm_label_ids` all -1, `dataset_name` / `dataset_type` ASSIGNED AS
ATTRIBUTES afterwards.  Dry run (tests/native_stub.py): the loss must come back under "random/test/masked_lm_loss"; that it is NaN when nothing
is to be predicted is arithmetic, asserted on hardware by tests/test_pretraining_gpu.py."""
import torch

from mmf_amd.common.sample import SampleList
from tests import golden_utils as G, model_utils as MU, native_stub


def test_pretrained_model_batch_form():
    z, case, cfg, sd, sample = G.load_pretraining_case()
    model = MU.build_visual_bert_pretraining(cfg, sd, device="cpu")
    T, R, D = sample["input_ids"].shape[1], 10, sample["image_feature_0"].shape[2]
    sample_list = SampleList()
    sample_list.add_field("input_ids", torch.randint(low=0, high=cfg["vocab_size"], size=(1, T)).long())
    sample_list.add_field("input_mask", torch.ones((1, T)).long())
    sample_list.add_field("segment_ids", torch.zeros(1, T).long())
    sample_list.add_field("image_feature_0", torch.rand((1, R, D)).float())
    sample_list.add_field("lm_label_ids", torch.zeros((1, T), dtype=torch.long).fill_(-1))
    model.eval()
    sample_list = sample_list.to("cpu")
    sample_list.dataset_name = "random"
    sample_list.dataset_type = "test"
    with native_stub.installed(), torch.no_grad():
        model_output = model(sample_list)
    assert "losses" in model_output
    assert "random/test/masked_lm_loss" in model_output["losses"]
    assert model_output["losses"]["random/test/masked_lm_loss"].numel() == 1
