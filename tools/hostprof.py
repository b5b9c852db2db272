"""Host-side profile of one train step (where does the Python/launch time go?)."""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
cfg = setup_cfg("configs/pt/final_c2f.yaml", ["MODEL.DEVICE", "cuda:0", "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 0])
torch.manual_seed(0)
tr = PTrainer(cfg)
g = torch.Generator().manual_seed(1)
data = tuple(synth_records(g, B, 800, 1333, 8, dev) for _ in range(4))
tr.run_step(data)
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.run_step(data)
torch.cuda.synchronize()
print("step ms", (time.perf_counter() - t0) * 1e3)
# how long does the host need if it never waits for the GPU?  (sync-free portions only show up as cumulative time)
pr = cProfile.Profile()
pr.enable()
tr.run_step(data)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
