"""LR schedules and optimiser options of the reference, as pure functions of the iteration.

Reference: pt/solver/build.py:22-57 (`build_lr_scheduler`: WarmupMultiStepLR, WarmupCosineLR [detectron2 0.5] and the
repo's own WarmupTwoStageMultiStepLR, pt/solver/lr_scheduler.py:22-66), driven by `hooks.LRScheduler`
(pt/engine/trainer.py:505).  The fused clip+SGD kernel takes the learning rate as a launch argument, so a schedule is
just `lr(it)`; there is no scheduler object to step.  torch.optim.SGD options the reference leaves at their defaults
(D2 build_optimizer: per-parameter groups with BIAS_LR_FACTOR / WEIGHT_DECAY_BIAS / WEIGHT_DECAY_NORM, NESTEROV) are
checked, not silently ignored."""
import bisect
import math


def warmup_factor_at_iter(method: str, it: int, warmup_iters: int, warmup_factor: float) -> float:
    """detectron2 `_get_warmup_factor_at_iter`."""
    if it >= warmup_iters:
        return 1.0
    if method == "constant":
        return warmup_factor
    if method == "linear":
        alpha = it / warmup_iters
        return warmup_factor * (1 - alpha) + alpha
    raise ValueError("Unknown warmup method: {}".format(method))


def lr_at(cfg, it: int) -> float:
    """Learning rate used by iteration `it` (== scheduler.get_lr() with last_epoch == it)."""
    S = cfg.SOLVER
    name = S.LR_SCHEDULER_NAME
    warm = warmup_factor_at_iter(S.WARMUP_METHOD, it, S.WARMUP_ITERS, S.WARMUP_FACTOR)
    if name == "WarmupMultiStepLR":
        return S.BASE_LR * warm * S.GAMMA ** bisect.bisect_right(list(S.STEPS), it)
    if name == "WarmupCosineLR":
        return S.BASE_LR * warm * 0.5 * (1.0 + math.cos(math.pi * it / S.MAX_ITER))
    if name == "WarmupTwoStageMultiStepLR":
        steps, factors = list(S.STEPS), list(S.FACTOR_LIST)
        if steps != sorted(steps):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(steps))
        if len(steps) + 1 != len(factors):
            raise ValueError("Length of milestones should match length of factor_list.")
        return S.BASE_LR * warm * factors[bisect.bisect_right(steps, it)]
    raise ValueError("Unknown LR scheduler: {}".format(name))


def check_optimizer_options(cfg) -> None:
    """The fused step implements torch.optim.SGD(momentum, weight_decay) with one setting for every parameter -- what the
    shipped configs select.  Anything else must fail loudly instead of training with the wrong optimiser."""
    S = cfg.SOLVER
    if S.NESTEROV:
        raise ValueError("SOLVER.NESTEROV=True is not implemented by the fused clip+SGD step")
    if float(S.BIAS_LR_FACTOR) != 1.0:
        raise ValueError("SOLVER.BIAS_LR_FACTOR != 1 is not implemented by the fused clip+SGD step")
    if float(S.WEIGHT_DECAY_BIAS) != float(S.WEIGHT_DECAY):
        raise ValueError("SOLVER.WEIGHT_DECAY_BIAS != SOLVER.WEIGHT_DECAY is not implemented by the fused clip+SGD step")
    lr_at(cfg, 0)          # unknown scheduler / warm-up names raise here, before the first step


def scheduler_state(cfg, last_iter: int, n_groups: int = 1) -> dict:
    """state_dict of the reference's `_LRScheduler` after finishing iteration `last_iter` (what fvcore's Checkpointer
    stores under "scheduler"); only `last_epoch` carries information."""
    nxt = last_iter + 1
    return {"last_epoch": nxt, "_step_count": nxt + 1, "base_lrs": [cfg.SOLVER.BASE_LR] * n_groups,
            "_last_lr": [lr_at(cfg, nxt)] * n_groups}
