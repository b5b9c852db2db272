"""Soak run: N full-size mutual-learning steps on rotating synthetic batches; prints step time, losses and allocator
state every 20 iterations (stability / memory-growth check)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_records  # noqa: E402
from probabilisticteacher_amd.config import setup_cfg  # noqa: E402
from probabilisticteacher_amd.engine import PTrainer  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 160
dev = torch.device("cuda:0")
cfg = setup_cfg("configs/pt/final_c2f.yaml", ["MODEL.DEVICE", "cuda:0", "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 0])
torch.manual_seed(0)
tr = PTrainer(cfg)
g = torch.Generator().manual_seed(99)
B = 16
batches = [tuple(synth_records(g, B, 800, 1333, 8, dev) for _ in range(4)) for _ in range(4)]
t0 = time.perf_counter()
for i in range(iters):
    m = tr.run_step(batches[i % 4])
    if i % 20 == 0 or i == iters - 1:
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (i + 1) * 1e3
        print(f"iter {i}: {ms:.0f} ms/it total_loss {m['total_loss']:.3f} grad_norm {m['grad_norm']:.1f} "
              f"allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB",
              flush=True)
        bad = {k: v for k, v in m.items() if v != v}
        if bad:                                  # which terms are NaN (an unsupervised term over ZERO rows is 0 / 0 in the reference too)
            print("   non-finite metrics:", sorted(bad), " all:", {k: round(v, 4) for k, v in m.items()}, flush=True)
