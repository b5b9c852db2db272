#!/usr/bin/env python
"""tools/exp/op_census.py [--amp] -- which source lines of the package launch the step's small device kernels: one bench step under
torch.profiler with Python stacks, kernel launches grouped by the innermost frame inside probabilisticteacher_amd/."""
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from probabilisticteacher_amd.config import setup_cfg  # noqa: E402
from probabilisticteacher_amd.engine import PTrainer  # noqa: E402

amp = "--amp" in sys.argv
dev = torch.device("cuda", 0)
B = 16
cfg = setup_cfg(os.path.join(bench.ROOT, "configs/pt/final_c2f.yaml"), [
    "MODEL.DEVICE", "cuda:0", "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 0, "SOLVER.IMG_PER_BATCH_LABEL", B,
    "SOLVER.IMG_PER_BATCH_UNLABEL", B, "SOLVER.AMP.ENABLED", amp])
K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
torch.manual_seed(0)
tr = PTrainer(cfg)
gen = torch.Generator().manual_seed(1234)
batches = [tuple(bench.synth_records(gen, B, 800, 1333, K, dev) for _ in range(4)) for _ in range(2)]
for i in range(2):
    tr.run_step(batches[i % 2])
torch.cuda.synchronize()
import traceback  # noqa: E402

from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

NOLAUNCH = ("view", "slice", "select", "as_strided", "unsqueeze", "squeeze", "expand", "reshape", "t", "transpose", "permute", "detach", "alias",
            "empty", "empty_like", "empty_strided", "_unsafe_view", "narrow", "unbind", "split", "lift_fresh", "_local_scalar_dense", "split_with_sizes",
            "is_pinned", "_pin_memory", "sym_size", "unfold", "_reshape_alias", "set_", "resize_", "new_empty", "new_empty_strided")
c = Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in NOLAUNCH:
            frame = "autograd / other"
            for fs in reversed(traceback.extract_stack(limit=25)):
                if "probabilisticteacher_amd" in fs.filename and not fs.filename.endswith(("ops.py", "_lib.py", "p8.py")):
                    frame = f"{fs.filename.split('probabilisticteacher_amd/')[-1]}:{fs.lineno}"
                    break
            else:
                for fs in reversed(traceback.extract_stack(limit=25)):
                    if "probabilisticteacher_amd" in fs.filename:
                        frame = f"{fs.filename.split('probabilisticteacher_amd/')[-1]}:{fs.lineno}"
                        break
            c[(frame, name)] += 1
        return func(*args, **(kwargs or {}))


with Census():
    tr.run_step(batches[0])
torch.cuda.synchronize()
print(sum(c.values()), "ATen operator calls (views excluded) in one step")
byframe = Counter()
for (f, n), v in c.items():
    byframe[f] += v
for f, v in byframe.most_common(60):
    ops_ = ", ".join(f"{n} x{k}" for (ff, n), k in c.most_common() if ff == f)
    print(f"{v:5d}  {f}   [{ops_[:140]}]")
