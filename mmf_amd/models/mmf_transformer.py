"""MMF Transformer (`mmft` / `mmf_transformer`) behind MMF's model API on the gfx950 kernels.

Mirrors mmf/models/mmf_transformer.py:33-445: same config keys (`modalities`, `backend`, `heads`, ...), the same
`preprocess_sample` contract (input / position / segment ids and masks per modality), `forward(sample_list) ->
{"scores": [B, num_labels]}`, and the reference's parameter tree (`backend.transformer.*`, `backend.embeddings.*`,
`encoders.*`, `heads.*`; no `model.` prefix).

Scope: modalities arrive as token ids (text) or pre-extracted feature rows (everything else) through identity
encoders — the raw-image CNN encoders of the reference are upstream of the hot path and raise here.
Heads: `mlp`.  The pretraining heads (mlm / itm / ...) are not built."""
import torch
from torch import nn

from mmf_amd.common.registry import registry
from mmf_amd.models.transformers.base import BaseTransformer
from mmf_amd.models.transformers.backends import huggingface as _hf_backend  # noqa: F401  (registers "huggingface")
from mmf_amd.models.transformers.heads import itm as _itm_head  # noqa: F401  (registers "itm")
from mmf_amd.models.transformers.heads import mlm as _mlm_head  # noqa: F401  (registers "mlm")
from mmf_amd.models.transformers.heads import mrc as _mrc_head  # noqa: F401  (registers "mrc")
from mmf_amd.models.transformers.heads import mrfr as _mrfr_head  # noqa: F401  (registers "mrfr")
from mmf_amd.models.transformers.heads import wra as _wra_head  # noqa: F401  (registers "wra")
from mmf_amd.models.transformers.heads import mlp as _mlp_head  # noqa: F401  (registers "mlp")


@registry.register_model("mmft")
@registry.register_model("mmf_transformer")
class MMFTransformer(BaseTransformer):
    def __init__(self, config, *args, **kwargs):
        super().__init__(config)
        self.modality_keys, self.modality_type, self.modality_segments = [], [], []
        for modality in self.config.modalities:
            self.modality_keys.append(modality["key"])
            self.modality_type.append(modality["type"])
            self.modality_segments.append(modality["segment_id"] if "segment_id" in modality else -1)

    @classmethod
    def format_state_key(cls, key):
        if key.startswith("pooler.") or key.startswith("classifier."):
            return key.replace("pooler.", "heads.0.pooler.").replace("classifier.", "heads.0.classifier.")
        return key

    @classmethod
    def config_path(cls):
        return "configs/models/mmf_transformer/defaults.yaml"

    def build_encoders(self):
        """mmf_transformer.py:113-146 — identity encoders only."""
        self.encoders = nn.ModuleDict()
        for modality in self.config.modalities:
            enc = modality.get("encoder", None)
            if enc is None and modality["type"] == "image" and "image_encoder" in self.config:
                enc = self.config.image_encoder
            kind = "identity" if enc is None else enc.get("type", "identity")
            if kind != "identity":
                raise NotImplementedError(
                    "modality %r: encoder type %r — only pre-extracted features through the identity encoder are on the "
                    "built path (SURVEY.md §8)" % (modality["key"], kind))
            self.encoders[modality["key"]] = nn.Identity()

    def tie_weights(self):
        """mmf_transformer.py:148-178: heads that tie to the text token embedding (`mlp` has nothing to tie)."""
        if "text" in self.modality_type:
            idx = self.modality_type.index("text")
            for head in self.heads:
                if self.config.get("tie_weight_to_encoder", None):
                    # :150-170: tie to the token embedding of a TRANSFORMER text encoder in front of the backend.  Only identity encoders
                    # are on the built path, and for those the reference raises the same error (no `transformer` / `embeddings` to tie to)
                    self._find_unique_encoder_key(self.config.tie_weight_to_encoder)
                    raise NotImplementedError("Current encoder module arch not supported.")
                head.tie_weights(self.backend.embeddings.token_embeddings[idx])

    def _find_unique_encoder_key(self, key):
        """mmf_transformer.py:433-445."""
        assert key in self.encoders, "MMFT doesn't have %s encoder." % key
        for modality in self.config.modalities:
            if modality["key"] == key:
                assert len([m for m in self.config.modalities if m["key"] == key]) == 1, "MMFT has multiple modalities with the same key %s." % key
                assert len([m for m in self.config.modalities if m["type"] == modality["type"]]) == 1, \
                    "Encoder %s should be the only encoder for %s." % (key, modality["type"])
                return key

    # ---- preprocess_sample, mmf_transformer.py:180-401 ------------------------------------------
    def preprocess_sample(self, sample_list):
        input_ids = self._infer_input_ids(sample_list)
        position_ids = self._infer_position_ids(input_ids)
        masks = self._infer_masks(sample_list, input_ids)
        segment_ids = self._infer_segment_ids(sample_list, input_ids)
        return {"input_ids": input_ids, "position_ids": position_ids, "segment_ids": segment_ids, "masks": masks,
                "mlm_labels": self._infer_mlm_labels(sample_list, input_ids),
                "itm_labels": self._infer_itm_labels(sample_list, input_ids)}

    def _check_keys_for_modality(self, sample_list, keys):
        assert len(keys) != 0
        for key in keys:
            if key in sample_list:
                return sample_list[key]
        expected = keys[0] if len(keys) == 1 else "%s or %s" % (", ".join(keys[:-1]), keys[-1])
        raise TypeError("Missing modality in SampleList. Expected to find %s" % expected)

    def _infer_input_ids(self, sample_list):
        input_ids, current_text_idx = {}, 0
        for idx, encoder in enumerate(self.encoders.values()):
            modality = self.modality_keys[idx]
            if self.modality_type[idx] == "text":
                text_ids = self._check_keys_for_modality(sample_list, ("input_ids", modality))
                if text_ids.dim() > 2:
                    input_ids[modality] = text_ids[:, current_text_idx]
                    current_text_idx += 1
                else:
                    input_ids[modality] = text_ids
            elif self.modality_type[idx] == "image":
                input_ids[modality] = self._check_keys_for_modality(sample_list, (modality, "image", "input_modal", "image_feature_0"))
            else:
                input_ids[modality] = self._check_keys_for_modality(sample_list, (modality,))
            if encoder is not None:
                input_ids[modality] = encoder(input_ids[modality])
            if self.modality_type[idx] != "text" and input_ids[modality].dim() == 2:
                input_ids[modality] = input_ids[modality].unsqueeze(1)
        return input_ids

    def _infer_position_ids(self, input_ids):
        # consecutive positions per modality (:277-289); the kernels take them as "row index + 0", so only the marker is kept
        return {m: torch.arange(0, input_ids[m].size(1), dtype=torch.long, device=input_ids[m].device).unsqueeze(0).expand(
            (input_ids[m].size(0), input_ids[m].size(1))) for m in self.modality_keys}

    def _infer_masks(self, sample_list, input_ids):
        masks, current_text_idx = {}, 0
        for idx, modality in enumerate(self.modality_keys):
            if self.modality_type[idx] == "text" and "input_mask" in sample_list:
                if sample_list["input_mask"].dim() > 2:
                    masks[modality] = sample_list["input_mask"][:, current_text_idx]
                    current_text_idx += 1
                else:
                    masks[modality] = sample_list["input_mask"]
            else:
                mask_attribute = "%s_mask" % modality
                if mask_attribute in sample_list:
                    masks[modality] = sample_list[mask_attribute]
                else:
                    masks[modality] = torch.ones(input_ids[modality].size()[:2], dtype=torch.long, device=input_ids[modality].device)
        return masks

    def _infer_segment_ids(self, sample_list, input_ids):
        segment_ids, current_text_idx = {}, 0
        for idx, modality in enumerate(self.modality_keys):
            if self.modality_segments[idx] == -1:
                continue
            if self.modality_type[idx] == "text" and "segment_ids" in sample_list:
                if sample_list["segment_ids"].dim() > 2:
                    segment_ids[modality] = sample_list["segment_ids"][:, current_text_idx]
                    current_text_idx += 1
                else:
                    segment_ids[modality] = sample_list["segment_ids"]
            else:
                segment_ids[modality] = torch.full(input_ids[modality].size()[:2], fill_value=self.modality_segments[idx],
                                                   dtype=torch.long, device=input_ids[modality].device)
        return segment_ids

    def _infer_itm_labels(self, sample_list, input_ids):
        if "is_correct" in sample_list:
            return {"is_correct": sample_list["is_correct"]}
        # `torch.tensor(True, dtype=torch.long)` of the reference (:364-373) as a fill kernel: a host scalar copied to the device is a synchronous
        # memcpy, which a hipGraph capture of the step refuses
        return {"is_correct": torch.ones((), dtype=torch.long, device=input_ids[self.modality_keys[0]].device)}

    def _infer_mlm_labels(self, sample_list, input_ids):
        mlm_labels, current_text_idx = {}, 0
        for idx, modality in enumerate(self.modality_keys):
            if self.modality_type[idx] == "text" and "lm_label_ids" in sample_list:
                if sample_list["lm_label_ids"].dim() > 2:
                    mlm_labels[modality] = sample_list["lm_label_ids"][:, current_text_idx]
                    current_text_idx += 1
                else:
                    mlm_labels[modality] = sample_list["lm_label_ids"]
            else:
                mlm_labels[modality] = torch.full(input_ids[modality].size()[:2], fill_value=-1, dtype=torch.long,
                                                  device=input_ids[modality].device)
        labels = [mlm_labels[m] for m in self.modality_keys]
        if labels:
            mlm_labels["combined_labels"] = torch.cat(labels, dim=-1)
        return mlm_labels

    def forward(self, sample_list):
        processed = self.preprocess_sample(sample_list)
        processed["target_key"] = sample_list
        masks = [processed["masks"][m] for m in self.modality_keys]
        sequence_output, encoded_layers = self.backend(processed["input_ids"], processed["position_ids"],
                                                       processed["segment_ids"], masks)
        return self.postprocess_output(sequence_output, encoded_layers, processed)

    def postprocess_output(self, sequence_output, encoded_layers, processed_sample_list):
        output_dict = {}
        for head in self.heads:
            output_dict.update(head(sequence_output, encoded_layers, processed_sample_list))
        return output_dict
