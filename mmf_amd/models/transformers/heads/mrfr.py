"""`MRFR` transformer head — masked region feature regression of UNITER (mmf/models/transformers/heads/mrfr.py:15-93) on the HIP kernels:
the masked regions are compacted (`compute_masked_hidden`, heads/utils.py:169-179 -> `functional.TakeRowsFn`), run through
Linear -> GELU -> LayerNorm (GEMM epilogue + LayerNorm kernel), projected back to feature space with the TIED image-embedding weight
applied transposed (`linear(h, W.t(), b)`, :86-88 — the GEMM reads W k-major, no transposed copy) and regressed onto the original
features with the mean squared error (:90; loss kernel fused with the projection's backward operands: `functional.TiedRegressionMSEFn`).
Parameter names as the reference's: `linear_proj_weight` (the very Parameter object of the image embedding), `linear_proj_bias`,
`feat_regress.{0,2}`."""
import torch
from torch import nn

from mmf_amd import functional as Fn
from mmf_amd.common.registry import registry
from mmf_amd.modules.hf_layers import LayerNorm, Linear


class _GeluSlot(nn.Module):
    """Place holder for `nn.GELU()` at index 1 of the reference's Sequential (no parameters; the GELU runs in the GEMM epilogue)."""


@registry.register_transformer_head("mrfr")
class MRFR(nn.Module):
    def __init__(self, img_embedding_weight, hidden_size=768, loss_name="mrfr_loss", mrfr_target_key="mrfr_region_target",
                 mrfr_mask_key="mrfr_region_mask", img_dim=2048, eps=1e-12, *args, **kwargs):
        super().__init__()
        self.loss_name = loss_name
        self.mrfr_target_key = mrfr_target_key
        self.mrfr_mask_key = mrfr_mask_key
        assert img_embedding_weight is not None and tuple(img_embedding_weight.shape) == (hidden_size, img_dim), (
            "MRFR head requires 'img_embedding_weight' with shape (%d, %d)." % (hidden_size, img_dim))
        self.linear_proj_weight = img_embedding_weight          # tied: the same Parameter as UNITERImageEmbeddings.img_linear.weight
        self.linear_proj_bias = nn.Parameter(torch.zeros(img_dim))
        self.feat_regress = nn.Sequential(Linear(hidden_size, hidden_size), _GeluSlot(), LayerNorm(hidden_size, eps=eps))
        nn.init.kaiming_uniform_(self.feat_regress[0].weight, a=5 ** 0.5)          # nn.Linear's default init scale (no checkpoint here)
        nn.init.zeros_(self.feat_regress[0].bias)

    def forward(self, sequence_output, processed_sample_list):
        output_dict = {}
        assert self.mrfr_target_key in processed_sample_list and processed_sample_list[self.mrfr_target_key] is not None, (
            "MRFR pretraining requires %s to be in sample list with value not None." % self.mrfr_target_key)
        feat_targets = processed_sample_list[self.mrfr_target_key]          # (n masked regions, img_dim)
        assert self.mrfr_mask_key in processed_sample_list and processed_sample_list[self.mrfr_mask_key] is not None, (
            "MRFR pretraining requires %s to be in sample list with value not None." % self.mrfr_mask_key)
        image_region_masks = processed_sample_list[self.mrfr_mask_key]      # (bs, num_feat) bool
        H = sequence_output.shape[-1]
        idx = image_region_masks.reshape(-1).nonzero().squeeze(1)           # (a host read-back, as the reference's boolean indexing)
        masked_output = Fn.TakeRowsFn.apply(sequence_output.reshape(-1, H), idx)
        dense, ln = self.feat_regress[0], self.feat_regress[2]
        hidden = ln(torch.ops.mmf_amd.dense_gelu(masked_output, dense.weight, dense.bias))
        loss = Fn.TiedRegressionMSEFn.apply(hidden, self.linear_proj_weight, self.linear_proj_bias, Fn.shadows.get(self.linear_proj_weight),
                                            feat_targets)
        output_dict["losses"] = {self.loss_name: loss}
        return output_dict
