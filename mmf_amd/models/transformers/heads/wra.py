"""`WRA` transformer head — word-region alignment of UNITER (mmf/models/transformers/heads/wra.py:14-83): the optimal-transport distance
between the text rows and the region rows of the joint sequence under the cosine cost, the transport plan approximated by 50 IPOT steps
(mmf/modules/ot.py), loss = (sum over matched pairs - sum over mismatched pairs) / number of pairs.  One HIP workgroup per sample keeps
the plan in LDS for all iterations (`functional.WordRegionAlignmentFn`, mmf_amd/csrc/uniter_ops.hip); fp32 arithmetic, as the reference
("run in fp32 for stability", wra.py:74).  No parameters."""
from torch import nn

from mmf_amd import functional as Fn
from mmf_amd.common.registry import registry


@registry.register_transformer_head("wra")
class WRA(nn.Module):
    def __init__(self, loss_name="wra_loss", ot_inputs_key="wra_info", wra_label_key="is_correct", *args, **kwargs):
        super().__init__()
        self.loss_name = loss_name
        self.ot_inputs_key = ot_inputs_key
        self.wra_label_key = wra_label_key

    def forward(self, sequence_output, processed_sample_list):
        output_dict = {}
        assert self.ot_inputs_key in processed_sample_list and processed_sample_list[self.ot_inputs_key] is not None, (
            "WRA pretraining requires %s to be in sample list with value not None." % self.ot_inputs_key)
        ot_inputs = processed_sample_list[self.ot_inputs_key]
        assert ot_inputs.get("txt_pad") is not None and ot_inputs.get("img_pad") is not None, (
            "WRA pretraining requires 'txt_pad', and 'img_pad' to be in 'processed_sample_list[%s]' with values not None." % self.ot_inputs_key)
        assert processed_sample_list.get(self.wra_label_key) is not None, (
            "WRA pretraining requires %s to be in sample list with value not None." % self.wra_label_key)
        tl = processed_sample_list["input_ids"].size(1)
        il = processed_sample_list["image_feat"].size(1)
        if tl > 128 or il > 128:
            raise NotImplementedError("WRA head: the IPOT kernel keeps the transport plan of one sample in LDS: at most 128 tokens and 128 "
                                      "regions, got %d and %d" % (tl, il))
        loss, _dist = Fn.WordRegionAlignmentFn.apply(sequence_output[:, :tl + il, :] if sequence_output.size(1) != tl + il else sequence_output,
                                                    tl, il, ot_inputs["txt_pad"], ot_inputs["img_pad"], processed_sample_list[self.wra_label_key])
        output_dict["losses"] = {self.loss_name: loss}
        return output_dict
