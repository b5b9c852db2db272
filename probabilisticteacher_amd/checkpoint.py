"""Teacher+student checkpoints in the reference's layout (reference
pt/checkpoint/detection_checkpoint.py:24-103, pt/modeling/meta_arch/ts_ensemble.py:20-29).

File format = what DetectionTSCheckpointer writes through fvcore's Checkpointer: a torch-saved dict with
"model" (keys `modelTeacher.*` / `modelStudent.*`), plus "optimizer"/"scheduler"/"iteration".  Here the optimiser
state is the flat momentum buffer of the fused clip+SGD step; a reference-written file (per-parameter torch SGD state)
is accepted for the model part and resumes with fresh momentum."""
import os
from typing import Dict

import torch


def save_checkpoint(trainer, path: str) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    model = {k: v.detach().cpu() for k, v in trainer.ensem_ts_model.state_dict().items()}
    torch.save({"model": model, "iteration": trainer.iter,
                "optimizer": {"flat_momentum": trainer.momentum_buf.detach().cpu(),
                              "first_step": trainer._first_step,
                              "layout": list(trainer.student.index.keys())}}, path)


def load_checkpoint(trainer, path: str, resume: bool = True) -> Dict:
    """Loads `modelTeacher.*`/`modelStudent.*`; a checkpoint that only has bare model keys (e.g. an ImageNet /
    burn-in student, detection_checkpoint.py:26-50) goes to the student only.  Returns the raw checkpoint dict."""
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt.get("model", ckpt)
    own = trainer.ensem_ts_model.state_dict()
    if any(k.startswith("modelStudent.") or k.startswith("modelTeacher.") for k in sd):
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError(f"checkpoint misses {len(missing)} keys, e.g. {missing[:3]}")
        with torch.no_grad():
            for k, v in own.items():
                v.copy_(sd[k])
    else:
        student = trainer.model.state_dict()
        with torch.no_grad():
            for k, v in student.items():
                if k in sd:
                    v.copy_(sd[k])
    if resume and "iteration" in ckpt:
        trainer.iter = trainer.start_iter = int(ckpt["iteration"])
        opt = ckpt.get("optimizer", {})
        if isinstance(opt, dict) and "flat_momentum" in opt and opt.get("layout") == list(trainer.student.index.keys()):
            trainer.momentum_buf.copy_(opt["flat_momentum"])
            trainer._first_step = bool(opt.get("first_step", False))
    return ckpt
