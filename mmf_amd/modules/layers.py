"""`ClassifierLayer` (mmf/modules/layers.py:99-123) for the classifier types on the built paths: `linear` (M4C's fixed
answer vocabulary, m4c.py:165-170 — the model reads `classifier.module.weight` as its answer embedding table) and `bert`
(BertPredictionHeadTransform + Linear, layers.py:126-160)."""
from torch import nn

from mmf_amd.modules.hf_layers import BertConfig, BertPredictionHeadTransform, Dropout, Linear


class ClassifierLayer(nn.Module):
    def __init__(self, classifier_type, in_dim, out_dim, **kwargs):
        super().__init__()
        if classifier_type == "linear":
            self.module = Linear(in_dim, out_dim)
            nn.init.kaiming_uniform_(self.module.weight, a=5 ** 0.5)      # nn.Linear's default init
            bound = 1.0 / in_dim ** 0.5
            nn.init.uniform_(self.module.bias, -bound, bound)
        elif classifier_type == "bert":
            config = kwargs.get("config", None) or BertConfig()
            if config.hidden_size != in_dim:
                raise ValueError("ClassifierLayer('bert'): in_dim must equal config.hidden_size")
            self.module = nn.Sequential(Dropout(config.hidden_dropout_prob), BertPredictionHeadTransform(config),
                                        Linear(in_dim, out_dim))
        else:
            raise NotImplementedError("Unknown / unbuilt classifier type: %s" % classifier_type)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
