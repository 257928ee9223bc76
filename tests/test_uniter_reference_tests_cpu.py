"""/root/reference/tests/models/test_uniter.py, ported: the same constructions, inputs and assertions (output shapes, loss names per task)
against `mmf_amd.models.uniter` — run as a host-logic dry run (tests/native_stub.py: every kernel wrapper is a shape / extent checker; there
is no CPU arithmetic path), so what is pinned here is what the reference's tests pin: the module tree accepts the reference's arguments, the
shapes come out right and every pretraining task reports its loss under the reference's name.  Numbers are the `-m gpu` tests' job."""
import pytest
import torch

from mmf_amd.common.sample import SampleList
from mmf_amd.models.uniter import UNITERForClassification, UNITERForPretraining, UNITERImageEmbeddings, UNITERModelBase
from tests import native_stub


def test_image_embeddings_forward():
    """test_uniter.py:21-37."""
    bs, num_feat = 32, 100
    config = {"img_dim": 1024, "hidden_size": 256, "pos_dim": 7}
    img_feat = torch.rand((bs, num_feat, config["img_dim"]))
    img_pos_feat = torch.rand((bs, num_feat, config["pos_dim"]))
    type_embeddings = torch.ones((bs, num_feat, 1), dtype=torch.long)
    embedding = UNITERImageEmbeddings(**config)
    with native_stub.installed():
        output = embedding(img_feat, img_pos_feat, type_embeddings, img_masks=None)
    assert list(output.shape) == [32, 100, 256]


def test_model_base_final_layer_shape():
    """test_uniter.py:40-66 (`test_pretrained_model`; the weights there come from the hub, here they are random: only the shape is asserted)."""
    img_dim = 1024
    model = UNITERModelBase(img_dim=img_dim, random_init=True)
    model.eval()
    bs, num_feats, max_sentence_len, pos_dim = 8, 100, 25, 7
    input_ids = torch.ones((bs, max_sentence_len), dtype=torch.long)
    img_feat = torch.rand((bs, num_feats, img_dim))
    img_pos_feat = torch.rand((bs, num_feats, pos_dim))
    position_ids = torch.arange(0, input_ids.size(1), dtype=torch.long).unsqueeze(0)
    attention_mask = torch.ones((bs, max_sentence_len + num_feats))
    with native_stub.installed(), torch.no_grad():
        model_output = model(input_ids, position_ids, img_feat, img_pos_feat, attention_mask).final_layer
    assert model_output.shape == torch.Size([8, 125, 768])


def _get_sample_list():
    """test_uniter.py:70-106."""
    bs, num_feats, max_sentence_len, img_dim, cls_dim = 8, 100, 25, 2048, 3129
    input_ids = torch.ones((bs, max_sentence_len), dtype=torch.long)
    input_mask = torch.ones((bs, max_sentence_len), dtype=torch.long)
    image_feat = torch.rand((bs, num_feats, img_dim))
    position_ids = torch.arange(0, max_sentence_len, dtype=torch.long).unsqueeze(0).expand(bs, -1)
    img_pos_feat = torch.rand((bs, num_feats, 7))
    attention_mask = torch.zeros((bs, max_sentence_len + num_feats), dtype=torch.long)
    image_mask = torch.zeros((bs, num_feats), dtype=torch.long)
    targets = torch.rand((bs, cls_dim))
    sample_list = SampleList()
    for k, v in (("input_ids", input_ids), ("input_mask", input_mask), ("image_feat", image_feat), ("img_pos_feat", img_pos_feat),
                 ("attention_mask", attention_mask), ("image_mask", image_mask), ("targets", targets), ("dataset_name", "test"),
                 ("dataset_type", "test"), ("position_ids", position_ids)):
        sample_list.add_field(k, v)
    return sample_list


def test_uniter_for_classification():
    """test_uniter.py:108-123."""
    heads = {"test": {"type": "mlp", "num_labels": 3129}}
    model = UNITERForClassification(head_configs=heads, loss_configs={"test": "logit_bce"}, tasks="test", random_init=True)
    model.eval()
    sample_list = _get_sample_list()
    with native_stub.installed(), torch.no_grad():
        model_output = model(sample_list)
    assert "losses" in model_output
    assert "test/test/logit_bce" in model_output["losses"]


def _enhance_sample_list_for_pretraining(sample_list):
    """test_uniter.py:125-141."""
    bs, sentence_len = sample_list["input_ids"].size(0), sample_list["input_ids"].size(1)
    num_feat, cls_dim = sample_list["image_feat"].size(1), 1601
    sample_list.add_field("is_correct", torch.ones((bs,), dtype=torch.long))
    sample_list.add_field("task", "mlm")
    sample_list.add_field("lm_label_ids", torch.zeros((bs, sentence_len), dtype=torch.long))
    sample_list.add_field("input_ids_masked", sample_list["input_ids"])
    sample_list.add_field("image_info_0", {"cls_prob": torch.rand((bs, num_feat, cls_dim))})


def test_uniter_for_pretraining():
    """test_uniter.py:143-180: one head per task, a forward pass through each, every loss under its reference name."""
    heads = {"mlm": {"type": "mlm"}, "itm": {"type": "itm"}, "mrc": {"type": "mrc"}, "mrfr": {"type": "mrfr"}, "wra": {"type": "wra"}}
    model = UNITERForPretraining(head_configs=heads, tasks="mlm,itm,mrc,mrfr,wra", mask_probability=0.15, random_init=True)
    model.eval()
    sample_list = _get_sample_list()
    _enhance_sample_list_for_pretraining(sample_list)
    expected_loss_names = {"mlm": "masked_lm_loss", "itm": "itm_loss", "mrc": "mrc_loss", "mrfr": "mrfr_loss", "wra": "wra_loss"}
    for task_name, loss_name in expected_loss_names.items():
        sample_list["task"] = task_name
        with native_stub.installed(), torch.no_grad():
            model_output = model(sample_list)
        assert "losses" in model_output
        assert loss_name in model_output["losses"], (task_name, list(model_output["losses"]))
