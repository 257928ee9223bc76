"""`MLP` transformer head (mmf/models/transformers/heads/mlp.py:22-94) on the HIP kernels: BertPooler (row 0 gather +
dense with the tanh in the GEMM epilogue) -> [dropout -> dense+GELU -> LayerNorm] x num_layers -> Linear."""
from torch import nn

from mmf_amd.common.registry import registry
from mmf_amd.models.transformers.base import BaseTransformerHead
from mmf_amd.modules.hf_layers import BertConfig, BertPooler, BertPredictionHeadTransform, Dropout, Linear


@registry.register_transformer_head("multilayer_mlp")
@registry.register_transformer_head("mlp")
class MLP(BaseTransformerHead):
    Config = dict(type="mlp", num_labels=2, hidden_size=768, hidden_dropout_prob=0.1, layer_norm_eps=1e-6, hidden_act="gelu",
                  pooler_name="bert_pooler", num_layers=1, in_dim=None)

    def __init__(self, config, *args, **kwargs):
        super().__init__(config, *args, **kwargs)
        self.num_labels = self.config.num_labels
        self.hidden_size = self.config.hidden_size
        self.in_dim = self.config.in_dim = self.hidden_size if self.config.in_dim is None else self.config.in_dim
        if self.config.hidden_act != "gelu":
            raise NotImplementedError("MLP head: only hidden_act='gelu' is fused in the GEMM epilogue")
        self.pooler = self.get_pooler(self.config.pooler_name)(BertConfig(hidden_size=self.in_dim))
        num_layers = self.config.get("num_layers", 1)
        assert num_layers >= 0
        layers, in_dim = [], self.in_dim
        tcfg = BertConfig(hidden_size=self.hidden_size, layer_norm_eps=self.config.layer_norm_eps)
        for _ in range(num_layers):
            layers.append(Dropout(self.config.hidden_dropout_prob))
            layers.append(BertPredictionHeadTransform(tcfg, in_dim=in_dim))
            in_dim = self.hidden_size
        self.classifier = nn.Sequential(*layers, Linear(self.hidden_size, self.num_labels))

    def forward(self, sequence_output, encoded_layers=None, processed_sample_list=None):
        assert sequence_output.size()[-1] == self.in_dim, "Mismatch between MLP head hidden_size and sequence_output last dim."
        x = self.pooler(sequence_output)
        for layer in list(self.classifier)[:-1]:
            x = layer(x)
        prediction = self.classifier[-1](x, out_f32=True)
        return {"scores": prediction.view(-1, self.num_labels)}

    def get_pooler(self, pooler_name):
        if pooler_name == "bert_pooler":
            return BertPooler
        if pooler_name == "identity":
            return lambda cfg: nn.Identity()
        raise NotImplementedError("%s is not implemented." % pooler_name)
