"""The learning-rate schedules the on-path project configs name (mmf/modules/schedulers.py):
`warmup_linear` (:34-37 -> transformers.get_linear_schedule_with_warmup; VisualBERT / ViLBERT / MMBT / UNITER / MMFT): linear warm-up from 0
to the base LR over `num_warmup_steps`, then linear decay to 0 at `num_training_steps`; `warmup_cosine` (:40-43 ->
get_cosine_schedule_with_warmup); `pythia` (:20-31, the default of `build_scheduler`; M4C) = `lr_lambda_update` of the global config's
`training.{use_warmup, warmup_iterations, warmup_factor, lr_steps, lr_ratio}` (mmf/utils/general.py:25-31); `multi_step` (:46-71), the same
rule from keyword arguments.  Host-side factors only: the fused AdamW reads the factor a scheduler sets (`group["lr"]`)."""
import math
from bisect import bisect, bisect_right

from torch.optim.lr_scheduler import LambdaLR

from mmf_amd.common.registry import registry


def lr_lambda_update(i_iter, cfg):
    """mmf/utils/general.py:25-31: linear ramp from `warmup_factor` to 1 over the warm-up iterations, then `lr_ratio` to the power of the
    number of `lr_steps` passed."""
    training = cfg["training"] if not hasattr(cfg, "training") else cfg.training
    if training["use_warmup"] is True and i_iter <= training["warmup_iterations"]:
        alpha = float(i_iter) / float(training["warmup_iterations"])
        return training["warmup_factor"] * (1.0 - alpha) + alpha
    return pow(training["lr_ratio"], bisect(list(training["lr_steps"]), i_iter))


@registry.register_scheduler("pythia")
class PythiaScheduler(LambdaLR):
    """schedulers.py:20-31: the factor comes from the GLOBAL configuration registered under "config"."""

    def __init__(self, optimizer, *args, **kwargs):
        self._lambda_func = lr_lambda_update
        self._global_config = registry.get("config")
        if self._global_config is None:
            raise RuntimeError("the 'pythia' scheduler reads training.{use_warmup, warmup_iterations, warmup_factor, lr_steps, lr_ratio} from the "
                               "configuration registered as registry.register('config', config): none is registered")
        super().__init__(optimizer, self.lr_lambda, *args, **kwargs)

    def lr_lambda(self, step):
        return self._lambda_func(step, self._global_config)


@registry.register_scheduler("multi_step")
class MultiStepScheduler(LambdaLR):
    """schedulers.py:46-71 (a PythiaScheduler whose `get_lr` is replaced; the global configuration is not consulted)."""

    def __init__(self, optimizer, *args, **kwargs):
        self.use_warmup = kwargs["use_warmup"]
        self.lr_steps = list(kwargs["lr_steps"])
        self.lr_ratio = kwargs["lr_ratio"]
        self.warmup_iterations = kwargs["warmup_iterations"] if self.use_warmup else 0
        self.warmup_factor = kwargs["warmup_factor"]
        assert self.warmup_iterations < self.lr_steps[0]
        super().__init__(optimizer, lambda step: 1.0)

    def get_lr(self):
        if self.last_epoch <= self.warmup_iterations and self.use_warmup is True:
            alpha = float(self.last_epoch) / float(self.warmup_iterations)
            lr_ratio = self.warmup_factor * (1.0 - alpha) + alpha
            return [base_lr * lr_ratio for base_lr in self.base_lrs]
        return [base_lr * self.lr_ratio ** bisect_right(self.lr_steps, self.last_epoch) for base_lr in self.base_lrs]


@registry.register_scheduler("warmup_cosine")
class WarmupCosineScheduler(LambdaLR):
    """schedulers.py:40-43 -> transformers.get_cosine_schedule_with_warmup: linear warm-up, then half a cosine period down to 0."""

    def __init__(self, optimizer, num_warmup_steps, num_training_steps, num_cycles=0.5, last_epoch=-1):
        def lr_lambda(current_step):
            if current_step < num_warmup_steps:
                return float(current_step) / float(max(1, num_warmup_steps))
            progress = float(current_step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))

        super().__init__(optimizer, lr_lambda, last_epoch)


@registry.register_scheduler("warmup_linear")
class WarmupLinearScheduler(LambdaLR):
    def __init__(self, optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
        def lr_lambda(current_step):
            if current_step < num_warmup_steps:
                return float(current_step) / float(max(1, num_warmup_steps))
            return max(0.0, float(num_training_steps - current_step) / float(max(1, num_training_steps - num_warmup_steps)))

        super().__init__(optimizer, lr_lambda, last_epoch)
