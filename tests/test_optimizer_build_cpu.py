"""The trainer-side construction of the optimizer step that follows the path: `build_optimizer` / `build_scheduler`
(mmf/utils/build.py:405-485), `get_optimizer_parameters` / `check_unused_parameters` / `clip_gradients` (mmf/utils/general.py:33-50, 136-180).
The two cases of /root/reference/tests/modules/test_optimizers.py ported (a plain model: one group; MMBT: the BERT recipe's two groups —
there built with `MMBT.from_params()`, which needs the network, here from the golden fixture's configuration), plus the error / warning
behaviours of the reference functions."""
import logging

import pytest
import torch
from torch import nn

from mmf_amd.models.base_model import BaseModel
from mmf_amd.utils.build import build_optimizer, build_scheduler
from mmf_amd.utils.configuration import Config
from mmf_amd.utils.general import check_unused_parameters, clip_gradients, get_optimizer_parameters
from tests import golden_utils as G, model_utils as MU


class SimpleModel(BaseModel):
    """tests/test_utils.py:195-223 of the reference, reduced to what the optimizer tests use."""

    def __init__(self, config):
        super().__init__(Config(dict(in_dim=1, out_dim=1), **dict(config)))

    def build(self):
        self.classifier = nn.Linear(self.config.in_dim, self.config.out_dim)

    def forward(self, batch):
        return {"scores": self.classifier(batch["x"])}


def _config():
    return Config(optimizer=dict(type="adam_w", params=dict(lr=5e-5)))


def test_build_optimizer_simple_model():
    model = SimpleModel({"in_dim": 1})
    model.build()
    optimizer = build_optimizer(model, _config())
    assert isinstance(optimizer, torch.optim.Optimizer)
    assert len(optimizer.param_groups) == 1
    assert optimizer.param_groups[0]["lr"] == 5e-5 and len(optimizer.param_groups[0]["params"]) == 2


def test_build_optimizer_custom_model():
    """MMBT hands its own groups over (`get_optimizer_parameters` -> the BERT recipe: decay / no decay)."""
    from oracle.mmbt_oracle import SHARED
    z, case, cfg, sd, sample = G.load_mmbt_case()
    model = MU.build_mmbt(cfg, sd, SHARED, device="cpu")
    config = _config()
    config["model"] = "mmbt"
    config["model_config"] = Config(mmbt=model.config)
    optimizer = build_optimizer(model, config)
    assert isinstance(optimizer, torch.optim.Optimizer)
    assert len(optimizer.param_groups) == 2
    assert sorted(g["weight_decay"] for g in optimizer.param_groups) == [0.0, 0.01]
    assert sum(len(g["params"]) for g in optimizer.param_groups) == len(list(model.parameters()))


def test_torch_optimizers_are_found_by_name_and_unknown_ones_refused():
    model = SimpleModel({})
    model.build()
    sgd = build_optimizer(model, Config(optimizer=dict(type="SGD", params=dict(lr=0.1))))
    assert type(sgd) is torch.optim.SGD
    with pytest.raises(ValueError, match="Optimizer attributes must have a 'type' key"):
        build_optimizer(model, Config(optimizer=dict(params=dict(lr=0.1))))
    with pytest.raises(ValueError, match="No optimizer class of type"):
        build_optimizer(model, Config(optimizer=dict(type="no_such_optimizer", params={})))
    with pytest.warns(UserWarning, match="optimizer attributes has no params defined"):
        build_optimizer(model, Config(optimizer=dict(type="Adam")))
    with pytest.raises(NotImplementedError, match="enable_state_sharding"):
        build_optimizer(model, Config(optimizer=dict(type="adam_w", params=dict(lr=1e-3), enable_state_sharding=True)))


def test_get_optimizer_parameters_forms():
    model = SimpleModel({})
    model.build()
    groups = get_optimizer_parameters(model, _config())
    assert isinstance(groups, list) and isinstance(groups[0], dict) and isinstance(groups[0]["params"], list)
    with pytest.raises(ValueError, match="optimizer got an empty parameter list"):
        get_optimizer_parameters(nn.Module(), _config())

    class Picky(SimpleModel):
        def build(self):
            super().build()
            self.extra = nn.Linear(1, 1)

        def get_optimizer_parameters(self, config):
            return (p for p in self.classifier.parameters())          # a generator of bare parameters, `extra` left out

    picky = Picky({})
    picky.build()
    groups = get_optimizer_parameters(picky, _config())
    assert len(groups) == 1 and len(groups[0]["params"]) == 2
    assert check_unused_parameters(groups, picky, _config()) == ["extra.weight", "extra.bias"]


def test_unused_parameters_are_logged(caplog):
    model = SimpleModel({})
    model.build()
    with caplog.at_level(logging.INFO, logger="mmf_amd.utils.general"):
        check_unused_parameters([{"params": [model.classifier.weight]}], model, _config())
    assert "Model parameters not used by optimizer: classifier.bias" in caplog.text


def test_build_scheduler():
    model = SimpleModel({})
    model.build()
    opt = build_optimizer(model, Config(optimizer=dict(type="SGD", params=dict(lr=1.0))))
    sched = build_scheduler(opt, Config(scheduler=dict(type="warmup_linear", params=dict(num_warmup_steps=2, num_training_steps=10))))
    lrs = []
    for _ in range(4):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    assert lrs == pytest.approx([0.0, 0.5, 1.0, 0.875])
    with pytest.raises(ValueError, match="No scheduler class of type"):
        build_scheduler(opt, Config(scheduler=dict(type="no_such_scheduler", params={})))
    from mmf_amd.common.registry import registry
    saved = registry.mapping["state"].pop("config", None)
    try:
        with pytest.warns(UserWarning, match="setting default to 'Pythia'"):
            with pytest.raises(RuntimeError):    # the default, 'pythia', needs the global configuration (none is registered here)
                build_scheduler(opt, Config())
    finally:
        if saved is not None:
            registry.register("config", saved)


def test_clip_gradients_modes():
    model = SimpleModel({})
    model.build()
    for p in model.parameters():
        p.grad = torch.full_like(p, 3.0)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    cfg = Config(training=dict(max_grad_l2_norm=None, clip_norm_mode="all"))
    assert clip_gradients(model, opt, 0, None, cfg) is None and float(model.classifier.weight.grad) == 3.0
    seen = {}

    class Writer:
        def add_scalars(self, d, i):
            seen.update(d, iteration=i)

    cfg = Config(training=dict(max_grad_l2_norm=1.0, clip_norm_mode="all"))
    norm = clip_gradients(model, opt, 7, Writer(), cfg, scale=2.0)          # torch.optim.SGD has no clip_grad_norm: nn.utils' is used
    assert float(norm) == pytest.approx((2 * 9.0) ** 0.5) and seen["iteration"] == 7 and float(seen["grad_norm"]) == float(norm)
    total = sum(float((p.grad ** 2).sum()) for p in model.parameters()) ** 0.5
    assert total == pytest.approx(2.0, rel=1e-4)                             # max_grad_l2_norm * scale

    class Own:
        def clip_grad_norm(self, max_norm):
            seen["asked"] = max_norm
            return torch.tensor(5.0)

    assert float(clip_gradients(model, Own(), 0, None, cfg)) == 5.0 and seen["asked"] == 1.0     # the optimizer's own reduction is preferred
    with pytest.raises(NotImplementedError, match="Clip norm mode question not implemented"):
        clip_gradients(model, opt, 0, None, Config(training=dict(max_grad_l2_norm=1.0, clip_norm_mode="question")))


def _sgd(lr=1.0):
    model = SimpleModel({})
    model.build()
    return torch.optim.SGD(model.parameters(), lr=lr)


def _run(opt, sched, n):
    out = []
    for _ in range(n):
        out.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    return out


def test_schedules_equal_the_transformers_functions():
    """`warmup_linear` / `warmup_cosine` are transformers' get_*_schedule_with_warmup in the reference (schedulers.py:34-43)."""
    import transformers
    from mmf_amd.common.registry import registry
    for name, ref_fn in (("warmup_linear", transformers.get_linear_schedule_with_warmup), ("warmup_cosine", transformers.get_cosine_schedule_with_warmup)):
        a, b = _sgd(0.3), _sgd(0.3)
        ours = registry.get_scheduler_class(name)(a, num_warmup_steps=3, num_training_steps=11)
        theirs = ref_fn(b, num_warmup_steps=3, num_training_steps=11)
        assert _run(a, ours, 13) == pytest.approx(_run(b, theirs, 13), abs=1e-12), name


def test_pythia_and_multi_step_schedules_follow_lr_lambda_update():
    """M4C's schedule (projects/m4c/configs/textvqa/defaults.yaml:76-86): warm-up from warmup_factor, then lr_ratio per lr_step passed."""
    from mmf_amd.common.registry import registry
    from mmf_amd.modules.schedulers import lr_lambda_update
    training = dict(use_warmup=True, warmup_iterations=4, warmup_factor=0.2, lr_steps=[6, 8], lr_ratio=0.1)
    cfg = Config(training=training)
    want = [0.2, 0.4, 0.6, 0.8, 1.0, 1.0, 0.1, 0.1, 0.01, 0.01]
    assert [lr_lambda_update(i, cfg) for i in range(10)] == pytest.approx(want)
    with pytest.raises(RuntimeError, match="'pythia' scheduler reads"):
        saved = registry.mapping["state"].pop("config", None)
        try:
            registry.get_scheduler_class("pythia")(_sgd())
        finally:
            if saved is not None:
                registry.register("config", saved)
    saved = registry.mapping["state"].get("config", None)
    registry.register("config", cfg)
    try:
        opt = _sgd(2.0)
        sched = build_scheduler(opt, Config(scheduler=dict(params={})))            # no type: the reference's default, 'pythia' (with its warning)
        assert _run(opt, sched, 10) == pytest.approx([2.0 * w for w in want])
    finally:
        if saved is None:
            registry.mapping["state"].pop("config", None)
        else:
            registry.register("config", saved)
    opt = _sgd(2.0)
    sched = registry.get_scheduler_class("multi_step")(opt, **training)
    assert _run(opt, sched, 10) == pytest.approx([2.0 * w for w in want])
    with pytest.raises(AssertionError):
        registry.get_scheduler_class("multi_step")(_sgd(), **dict(training, warmup_iterations=7))
    assert lr_lambda_update(100, Config(training=dict(training, use_warmup=False))) == pytest.approx(0.01)
