"""How much host CPU time does one train step need (thread CPU time vs wall time), and how many host<->device syncs?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_records  # noqa: E402
from probabilisticteacher_amd.config import setup_cfg  # noqa: E402
from probabilisticteacher_amd.engine import PTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
cfg = setup_cfg("configs/pt/final_c2f.yaml", ["MODEL.DEVICE", "cuda:0", "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 0])
torch.manual_seed(0)
tr = PTrainer(cfg)
g = torch.Generator().manual_seed(1)
data = tuple(synth_records(g, B, 800, 1333, 8, dev) for _ in range(4))
for _ in range(2):
    tr.run_step(data)
torch.cuda.synchronize()
for _ in range(3):
    w0, c0 = time.perf_counter(), time.thread_time()
    tr.run_step(data)
    c1 = time.thread_time()
    torch.cuda.synchronize()
    w1 = time.perf_counter()
    print(f"step wall {1e3 * (w1 - w0):.1f} ms, main-thread CPU {1e3 * (c1 - c0):.1f} ms, load {os.getloadavg()[0]:.1f}")
