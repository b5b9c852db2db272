#!/usr/bin/env python
"""tools/exp/host_bound.py [--amp] -- how much of a step is the GPU waiting for the host?  Runs the bench's step (a) as is (the
metrics read-back synchronises every step) and (b) with the read-back stubbed out, so that the host may run ahead across steps:
(b)'s time per step is the GPU-bound step time, the host loop's own time per step the enqueue cost."""
import os
import sys
import time

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
pool = torch.empty(48 << 30, dtype=torch.uint8, device=dev)
del pool
for i in range(3):
    tr.run_step(batches[i % 2])
torch.cuda.synchronize()


def run(n):
    t0 = time.perf_counter()
    host = []
    for i in range(n):
        ts = time.perf_counter()
        tr.run_step(batches[i % 2])
        host.append(1e3 * (time.perf_counter() - ts))
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, host


ms, host = run(8)
print(f"(a) as is: {ms:.1f} ms per step")
orig = PTrainer._write_metrics
PTrainer._write_metrics = lambda self, rd, dt, ss: setattr(self, "last_metrics", {})
ms, host = run(8)
print(f"(b) no read-back: {ms:.1f} ms per step; host loop per step {[round(h, 1) for h in host]}")
