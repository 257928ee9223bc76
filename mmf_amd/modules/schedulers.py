"""`warmup_linear` (mmf/modules/schedulers.py:34-37 -> transformers.get_linear_schedule_with_warmup): linear
warm-up from 0 to the base LR over `num_warmup_steps`, then linear decay to 0 at `num_training_steps`."""
from torch.optim.lr_scheduler import LambdaLR

from mmf_amd.common.registry import registry


@registry.register_scheduler("warmup_linear")
class WarmupLinearScheduler(LambdaLR):
    def __init__(self, optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
        def lr_lambda(current_step):
            if current_step < num_warmup_steps:
                return float(current_step) / float(max(1, num_warmup_steps))
            return max(0.0, float(num_training_steps - current_step) / float(max(1, num_training_steps - num_warmup_steps)))

        super().__init__(optimizer, lr_lambda, last_epoch)
