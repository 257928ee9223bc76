"""Multi-task head container of the transformer-head registry (mmf/models/transformers/heads/utils.py): `build_heads_dict` turns the
`heads` section of a model config — one list of head configs, or a mapping task -> head config(s) — into a `HeadsDict`, whose forward runs
the heads of one task on the encoder output, applies each head's `MMFLoss` where the head returns scores instead of losses, and merges
what they return.  Host-side Python like the reference (the heads themselves run on the HIP kernels: heads/{mlp,itm,mlm,mrc,mrfr,wra}.py).
`compute_masked_hidden` (utils.py:150-162) is the boolean row selection the masked-region heads use in the reference; the heads of this
package select rows with the index-compaction kernel instead (`functional.TakeRowsFn`), the function is kept for callers of the API."""
import collections.abc

from torch import nn

from mmf_amd.common.registry import registry


def _make_head(conf):
    return registry.get_transformer_head_class(conf.get("type", "mlp"))(conf)


def _describe(confs):
    """(heads, loss names, head type names) of a list of head configs."""
    return (nn.ModuleList([_make_head(c) for c in confs]), [c.get("loss") for c in confs], [c.get("type", "mlp") for c in confs])


def build_heads_dict(head_configs, tasks, losses):
    """utils.py:11-66.  A sequence of head configs: every head runs for every batch (task = None).  A mapping: for each name in `tasks` its
    entry — one config or a list of them — becomes that task's head list; a task without an entry is an error.  `losses`: name -> MMFLoss,
    looked up through each head config's `loss` key."""
    if isinstance(head_configs, collections.abc.Mapping):
        heads, loss_names, head_names = nn.ModuleDict(), {}, {}
        for task in tasks:
            conf = head_configs.get(task)
            if conf is None:
                raise ValueError("No head defined for %s. Dataset task %s requires a head to return dict with 'losses'" % (task, task))
            confs = conf if isinstance(conf, collections.abc.Sequence) else [conf]
            heads[task], loss_names[task], head_names[task] = _describe(confs)
        return HeadsDict(heads, head_names, losses, loss_names)
    if isinstance(head_configs, collections.abc.Sequence):
        heads, loss_names, head_names = _describe(list(head_configs))
        return HeadsDict(heads, head_names, losses, loss_names)
    raise TypeError("heads must be a list of head configs or a mapping task -> head config(s), got %s" % type(head_configs).__name__)


class HeadsDict(nn.Module):
    """utils.py:69-147: the heads of a multi-task transformer model and the losses that go with them."""

    def __init__(self, heads, head_names, losses, head_loss_names):
        super().__init__()
        self.heads = heads
        self.head_names = head_names
        self.losses = losses
        self.head_loss_names = head_loss_names

    def _of(self, task):
        if isinstance(self.heads, nn.ModuleList):
            return self.heads, self.head_loss_names, self.head_names
        return self.heads[task], self.head_loss_names[task], self.head_names[task]

    def forward(self, task, sequence, sample_list):
        """Every head of `task` on `sequence`; losses under the same key are added up, `scores` is the sum of the heads' scores (0 when no
        head returns any)."""
        heads, loss_names, names = self._of(task)
        assert len(loss_names) == len(heads)
        merged, scores = {}, 0
        for head, loss_name, name in zip(heads, loss_names, names):
            result = self._process_head_output(head(sequence, processed_sample_list=sample_list), loss_name, name, sample_list)
            for key, value in result["losses"].items():
                merged[key] = merged[key] + value if key in merged else value
            scores = scores + result.get("scores", 0)
        return {"losses": merged, "scores": scores}

    def _process_head_output(self, outputs, loss_name, head_name, sample_list):
        """A head that already returns `losses` is taken as it is; otherwise its scores (a tensor, or the `scores` entry of a dict) are
        flattened to rows and handed to the MMFLoss its config names."""
        is_dict = isinstance(outputs, collections.abc.MutableMapping)
        if is_dict and "losses" in outputs:
            return outputs
        logits = outputs["scores"] if (is_dict and "scores" in outputs) else outputs
        logits = logits.contiguous().view(-1, logits.size(-1))
        if loss_name is None:
            raise ValueError("Transformer head %s must either define a 'loss' in its config or return a dict that contains key 'losses'." % head_name)
        return {"losses": self.losses[loss_name](sample_list, {"scores": logits}), "scores": logits}


def compute_masked_hidden(hidden, mask):
    """Rows of `hidden` [B, N, D] where the boolean `mask` [B, N] is set, as [count, D] (utils.py:150-162)."""
    return hidden[mask.unsqueeze(-1).expand_as(hidden)].contiguous().view(-1, hidden.size(-1))
