import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import synth_records
from probabilisticteacher_amd.config import setup_cfg
from probabilisticteacher_amd.engine import PTrainer
dev = torch.device("cuda:0")
cfg = setup_cfg("configs/pt/final_c2f.yaml", ["MODEL.DEVICE", "cuda:0", "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 0])
torch.manual_seed(0)
tr = PTrainer(cfg)
g = torch.Generator().manual_seed(1234)
B = 16
batches = [tuple(synth_records(g, B, 800, 1333, 8, dev) for _ in range(4)) for _ in range(2)]
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.run_step(batches[i % 2])
    torch.cuda.synchronize()
    print(f"step {i} batch {i % 2}: {1e3 * (time.perf_counter() - t0):.1f} ms, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB, "
          f"alloc retries {torch.cuda.memory_stats().get('num_alloc_retries', 0)}, segments {torch.cuda.memory_stats().get('segment.all.allocated', 0)}")
