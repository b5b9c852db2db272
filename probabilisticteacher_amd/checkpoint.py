"""Teacher+student checkpoints in the reference's on-disk format.

Reference: pt/checkpoint/detection_checkpoint.py:24-103 (`DetectionTSCheckpointer`), pt/modeling/meta_arch/
ts_ensemble.py:20-29 (`EnsembleTSModel` key prefixes), pt/engine/trainer.py:466-496 (`resume_or_load`), fvcore's
Checkpointer / PeriodicCheckpointer (file layout, `last_checkpoint` tag file, `model_{iter:07d}.pth`, `model_final.pth`).

File = torch-saved dict:
    "model"      `modelTeacher.*` / `modelStudent.*` tensors (EnsembleTSModel.state_dict())
    "optimizer"  torch.optim.SGD.state_dict() as D2's build_optimizer lays it out: ONE param group per trainable parameter
                 in `named_parameters()` order, state[i]["momentum_buffer"] -- written from / read into the flat momentum
                 buffer of the fused clip+SGD step, so files move between the reference and this build in both directions
    "scheduler"  _LRScheduler.state_dict() (last_epoch)
    "iteration"  the iteration that just FINISHED; resuming starts at iteration + 1 (trainer.py:490-493)"""
import os
from collections import namedtuple
from typing import Dict, List, Optional

import torch

from .solver import lr_at, scheduler_state

IncompatibleKeys = namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys", "incorrect_shapes"])


def _trainable_names(trainer) -> List[str]:
    """D2 get_default_optimizer_params order: modules() pre-order == named_parameters() order, trainable only."""
    return [n for n, p in trainer.model.named_parameters() if p.requires_grad]


def optimizer_state_dict(trainer) -> Dict:
    """The flat momentum buffer as a torch.optim.SGD state_dict with one group per parameter."""
    S = trainer.cfg.SOLVER
    names = _trainable_names(trainer)
    # the scheduler has already stepped when a checkpoint is written (hooks.LRScheduler.after_step precedes the checkpointer):
    # the file holds the learning rate of the NEXT iteration to run
    lr = lr_at(trainer.cfg, trainer.iter)
    groups, state = [], {}
    for i, n in enumerate(names):
        groups.append({"lr": lr, "momentum": S.MOMENTUM, "dampening": 0, "weight_decay": S.WEIGHT_DECAY,
                       "nesterov": bool(S.NESTEROV), "initial_lr": S.BASE_LR, "params": [i]})
        if not trainer._first_step:                    # torch creates momentum_buffer at the first step()
            off, k = trainer.student.index[n]
            state[i] = {"momentum_buffer": trainer.momentum_buf[off:off + k].view(trainer.student.params[n].shape)
                        .detach().cpu().clone()}
    return {"state": state, "param_groups": groups}


def load_optimizer_state_dict(trainer, sd: Dict) -> None:
    names = _trainable_names(trainer)
    order = [i for g in sd.get("param_groups", []) for i in g["params"]]
    if len(order) != len(names):
        raise ValueError(f"optimizer state has {len(order)} parameters, the model has {len(names)} trainable ones")
    state = sd.get("state", {})
    trainer.momentum_buf.zero_()
    have = 0
    for pos, n in zip(order, names):
        st = state.get(pos, state.get(str(pos)))
        if st is None or st.get("momentum_buffer") is None:
            continue
        off, k = trainer.student.index[n]
        buf = st["momentum_buffer"]
        if buf.numel() != k:
            raise ValueError(f"momentum buffer of {n}: {tuple(buf.shape)} does not fit {k} elements")
        trainer.momentum_buf[off:off + k].copy_(buf.reshape(-1))
        have += 1
    trainer._first_step = have == 0


def _load_into(module_sd: Dict[str, torch.Tensor], src: Dict[str, torch.Tensor], prefix: str = "") -> IncompatibleKeys:
    """Non-strict load with fvcore's shape check (detection_checkpoint.py:84-103): mismatching shapes are reported
    and skipped, never broadcast."""
    missing, bad, used = [], [], set()
    with torch.no_grad():
        for k, v in module_sd.items():
            sk = prefix + k
            if sk not in src:
                missing.append(sk)
                continue
            used.add(sk)
            t = src[sk]
            if not torch.is_tensor(t):
                t = torch.as_tensor(t)                               # _convert_ndarray_to_tensor
            if tuple(t.shape) != tuple(v.shape):
                bad.append((sk, tuple(t.shape), tuple(v.shape)))
                continue
            v.copy_(t)
    unexpected = [k for k in src if k not in used and (not prefix or k.startswith(prefix))]
    return IncompatibleKeys(missing, unexpected, bad)


def load_model(trainer, checkpoint: Dict) -> IncompatibleKeys:
    """DetectionTSCheckpointer._load_model: a Caffe2-tagged file (ImageNet / converted weights) updates the student
    only (:26-50); otherwise the whole EnsembleTSModel (:52-73).  A file with bare (un-prefixed) model keys -- a plain
    GuassianGeneralizedRCNN state_dict, e.g. a burn-in student saved outside the ensemble -- also goes to the student."""
    sd = dict(checkpoint["model"]) if "model" in checkpoint else dict(checkpoint)
    if any(k.startswith("module.") for k in sd):                     # _strip_prefix_if_present(..., "module.")
        if all(k.startswith("module.") for k in sd):
            sd = {k[len("module."):]: v for k, v in sd.items()}
    ensemble = any(k.startswith("modelStudent.") or k.startswith("modelTeacher.") for k in sd)
    if checkpoint.get("__author__", None) == "Caffe2" or not ensemble:
        return _load_into(trainer.model.state_dict(), sd)
    return _load_into(trainer.ensem_ts_model.state_dict(), sd)


def save_checkpoint(trainer, path: str, iteration: Optional[int] = None, tag_last: bool = True) -> str:
    """fvcore Checkpointer.save: `iteration` = the iteration that just finished (default: trainer.iter - 1)."""
    it = trainer.iter - 1 if iteration is None else int(iteration)
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    model = {k: v.detach().cpu() for k, v in trainer.ensem_ts_model.state_dict().items()}
    osd = optimizer_state_dict(trainer)
    torch.save({"model": model, "optimizer": osd, "scheduler": scheduler_state(trainer.cfg, it, len(osd["param_groups"])),
                "iteration": it}, path)
    if tag_last:
        with open(os.path.join(d, "last_checkpoint"), "w") as f:
            f.write(os.path.basename(path))
    return path


def load_checkpoint(trainer, path: str, resume: bool = True) -> IncompatibleKeys:
    """trainer.py:466-496: resume=True also restores the optimiser state and continues at iteration + 1; resume=False
    loads weights only and starts at iteration 0.  Missing / unexpected / wrongly-shaped keys are returned and, for a
    resume, missing model keys are an error."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    inc = load_model(trainer, ckpt)
    if resume:
        if inc.missing_keys or inc.incorrect_shapes:
            raise KeyError(f"cannot resume from {path}: missing {inc.missing_keys[:3]} wrong shapes {inc.incorrect_shapes[:3]}")
        if isinstance(ckpt.get("optimizer"), dict) and "param_groups" in ckpt["optimizer"]:
            load_optimizer_state_dict(trainer, ckpt["optimizer"])
        trainer.iter = trainer.start_iter = int(ckpt.get("iteration", -1)) + 1
    return inc


def last_checkpoint(output_dir: str) -> Optional[str]:
    tag = os.path.join(output_dir, "last_checkpoint")
    if not os.path.exists(tag):
        return None
    with open(tag) as f:
        return os.path.join(output_dir, f.read().strip())


def resume_or_load(trainer, resume: bool = False) -> Optional[IncompatibleKeys]:
    """PTrainer.resume_or_load (trainer.py:466-496): the file is cfg.MODEL.WEIGHTS; with resume=True and no
    MODEL.WEIGHTS the run continues from OUTPUT_DIR's `last_checkpoint` (fvcore's resume behaviour)."""
    path = trainer.cfg.MODEL.WEIGHTS
    if resume and not path:
        path = last_checkpoint(trainer.cfg.OUTPUT_DIR) or ""
    if not path:
        return None
    return load_checkpoint(trainer, path, resume=resume)


class PeriodicCheckpointer:
    """fvcore PeriodicCheckpointer as hooks.PeriodicCheckpointer drives it (trainer.py:521-527): after iteration `it`,
    `model_{it:07d}.pth` every `period` iterations and `model_final.pth` after the last one; rank 0 only."""

    def __init__(self, trainer, period: int, max_iter: int, output_dir: Optional[str] = None):
        self.trainer, self.period, self.max_iter = trainer, int(period), int(max_iter)
        self.dir = output_dir or trainer.cfg.OUTPUT_DIR

    def step(self, it: int) -> Optional[str]:
        out = None
        if self.period > 0 and (it + 1) % self.period == 0:
            out = save_checkpoint(self.trainer, os.path.join(self.dir, "model_{:07d}.pth".format(it)), iteration=it)
        if it >= self.max_iter - 1:
            out = save_checkpoint(self.trainer, os.path.join(self.dir, "model_final.pth"), iteration=it)
        return out
