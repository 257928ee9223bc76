"""`ITM` transformer head (mmf/models/transformers/heads/itm.py:19-74) on the HIP kernels: BertPooler (row 0, dense + tanh in the
GEMM epilogue) -> HF BertOnlyNSPHead (`seq_relationship`: Linear(hidden, 2)) -> CrossEntropyLoss(ignore_index) against
`itm_labels.is_correct`."""
from torch import nn

from mmf_amd import functional as Fn
from mmf_amd.common.registry import registry
from mmf_amd.models.transformers.base import BaseTransformerHead
from mmf_amd.modules.hf_layers import BertConfig, BertPooler, Linear

LABEL_KEY = "itm_labels"


class BertOnlyNSPHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.seq_relationship = Linear(config.hidden_size, 2)


@registry.register_transformer_head("itm")
class ITM(BaseTransformerHead):
    Config = dict(type="itm", hidden_size=768, loss_name="itm_loss", ignore_index=-1, itm_label_key="is_correct")

    def __init__(self, config, *args, **kwargs):
        super().__init__(config, *args, **kwargs)
        cfg = BertConfig(hidden_size=self.config.hidden_size)
        self.pooler = BertPooler(cfg)
        self.cls = BertOnlyNSPHead(cfg)

    def forward(self, sequence_output, encoded_layers=None, processed_sample_list=None):
        assert processed_sample_list is not None, "ITM head requires 'processed_sample_list' argument"
        output_dict = {}
        if self.config.itm_label_key in processed_sample_list:
            next_sentence_labels = processed_sample_list[self.config.itm_label_key]
        else:
            assert LABEL_KEY in processed_sample_list and processed_sample_list[LABEL_KEY] is not None, (
                "ITM pretraining requires %s to be in sample list with value not None." % LABEL_KEY)
            next_sentence_labels = processed_sample_list[LABEL_KEY][self.config.itm_label_key]
        pooled_output = self.pooler(sequence_output)
        seq_relationship_score = self.cls.seq_relationship(pooled_output, out_f32=True)
        itm_loss = Fn.CrossEntropyFn.apply(seq_relationship_score.contiguous().view(-1, 2),
                                           next_sentence_labels.contiguous().view(-1).long(), int(self.config.ignore_index))
        output_dict["losses"] = {self.config.loss_name: itm_loss}
        return output_dict
