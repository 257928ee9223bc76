"""`MRC` transformer head — masked region classification of UNITER (mmf/models/transformers/heads/mrc.py:15-90) on the HIP kernels: the
masked regions are compacted (`compute_masked_hidden`, heads/utils.py:169-179 -> `functional.TakeRowsFn`), run through
Linear -> GELU -> LayerNorm (GEMM epilogue + LayerNorm kernel) and the region-class decoder fused with its loss: KLDivLoss(batchmean)
against the detector's soft labels (`use_kl`, the soft-target KL kernels) or cross-entropy against their argmax over the
non-background classes (the vocabulary cross-entropy kernels).  Parameter names as the reference's `region_classifier.{0,2,3}`."""
import torch
from torch import nn

from mmf_amd import functional as Fn
from mmf_amd.common.registry import registry
from mmf_amd.modules.hf_layers import LayerNorm, Linear


class _GeluSlot(nn.Module):
    """Place holder for `nn.GELU()` at index 1 of the reference's Sequential (no parameters; the GELU runs in the GEMM epilogue)."""


@registry.register_transformer_head("mrc")
class MRC(nn.Module):
    def __init__(self, hidden_size=768, loss_name="mrc_loss", ignore_index=-1, mrc_label_key="region_class",
                 mrc_mask_key="image_region_mask", label_dim=1601, eps=1e-12, use_kl=True, *args, **kwargs):
        super().__init__()
        self.loss_name = loss_name
        self.ignore_index = ignore_index
        self.mrc_label_key = mrc_label_key
        self.mrc_mask_key = mrc_mask_key
        self.use_kl = use_kl
        self.region_classifier = nn.Sequential(Linear(hidden_size, hidden_size), _GeluSlot(), LayerNorm(hidden_size, eps=eps),
                                               Linear(hidden_size, label_dim))
        for m in (self.region_classifier[0], self.region_classifier[3]):      # nn.Linear's default init scale (no checkpoint here)
            nn.init.kaiming_uniform_(m.weight, a=5 ** 0.5)
            nn.init.zeros_(m.bias)

    def forward(self, sequence_output, processed_sample_list):
        output_dict = {}
        assert self.mrc_label_key in processed_sample_list and processed_sample_list[self.mrc_label_key] is not None, (
            "MRC pretraining requires %s to be in sample list with value not None." % self.mrc_label_key)
        region_labels = processed_sample_list[self.mrc_label_key]          # (n masked regions, label_dim)
        assert self.mrc_mask_key in processed_sample_list and processed_sample_list[self.mrc_mask_key] is not None, (
            "MRC pretraining requires %s to be in sample list with value not None." % self.mrc_mask_key)
        image_region_masks = processed_sample_list[self.mrc_mask_key]      # (bs, num_feat) bool
        H = sequence_output.shape[-1]
        idx = image_region_masks.reshape(-1).nonzero().squeeze(1)          # (a host read-back, as the reference's boolean indexing)
        masked_output = Fn.TakeRowsFn.apply(sequence_output.reshape(-1, H), idx)
        dense, ln, dec = self.region_classifier[0], self.region_classifier[2], self.region_classifier[3]
        hidden = ln(torch.ops.mmf_amd.dense_gelu(masked_output, dense.weight, dense.bias))
        n = idx.numel()
        if self.use_kl:
            ones = torch.ones(n, dtype=torch.int64, device=hidden.device)
            loss, _ = torch.ops.mmf_amd.masked_region_head(hidden, dec.weight, dec.bias, region_labels, ones)     # sum / n = batchmean
        else:
            label_targets = torch.max(region_labels[:, 1:], dim=-1)[1] + 1          # background class should not be the target
            loss, _ = torch.ops.mmf_amd.masked_lm_head(hidden, dec.weight, dec.bias, label_targets, int(self.ignore_index))
        output_dict["losses"] = {self.loss_name: loss}
        return output_dict
