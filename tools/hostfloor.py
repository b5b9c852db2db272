"""Host floor of one train step: the same batch structure (16 + 16 records) at a tiny image size, where the GPU work is
negligible and the wall time is the Python + launch time the host needs per step; with cProfile on top."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_records  # noqa: E402
from probabilisticteacher_amd.config import setup_cfg  # noqa: E402
from probabilisticteacher_amd.engine import PTrainer  # noqa: E402

B = 16
dev = torch.device("cuda:0")
cfg = setup_cfg("configs/pt/final_c2f.yaml", ["MODEL.DEVICE", "cuda:0", "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 0])
torch.manual_seed(0)
tr = PTrainer(cfg)
g = torch.Generator().manual_seed(1)
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (96, 128)
data = tuple(synth_records(g, B, H, W, 8, dev, m=3) for _ in range(4))
for _ in range(3):
    tr.run_step(data)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    tr.run_step(data)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
print(f"{W}x{H}: step wall ms {['%.1f' % t for t in ts]}")
pr = cProfile.Profile()
pr.enable()
tr.run_step(data)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(28)
