"""List the device->host synchronisation points of one train step (call site, count, time spent waiting)."""
import collections
import os
import sys
import time
import traceback

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
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 1333)
data = tuple(synth_records(g, B, H, W, 8, dev) for _ in range(4))
for _ in range(2):
    tr.run_step(data)
torch.cuda.synchronize()
log = collections.OrderedDict()


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        t0 = time.perf_counter()
        r = orig(*a, **k)
        dt = time.perf_counter() - t0
        is_dev = any(isinstance(x, torch.Tensor) and x.is_cuda for x in a)
        if is_dev:
            st = [s for s in traceback.extract_stack()[:-1] if "probabilisticteacher_amd" in s.filename]
            site = f"{os.path.basename(st[-1].filename)}:{st[-1].lineno} {name}" if st else name
            c = log.setdefault(site, [0, 0.0])
            c[0] += 1
            c[1] += dt
        return r
    setattr(owner, name, f)


for nm in ("cpu", "item", "tolist", "nonzero"):
    wrap(torch.Tensor, nm)
wrap(torch, "nonzero")
t0 = time.perf_counter()
tr.run_step(data)
torch.cuda.synchronize()
print(f"step {1e3 * (time.perf_counter() - t0):.1f} ms")
for k, (n, t) in log.items():
    print(f"{n:4d} x {1e3 * t:8.2f} ms  {k}")
