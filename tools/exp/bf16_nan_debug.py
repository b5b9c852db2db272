"""Experiment: when the native-bf16 run reports a NaN loss term, replay the SAME step (same parameters, batch, sampler keys,
ratios) in bf16_emulate mode and print both."""
import os, random, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from loss_curve_parity import make_batch
from oracle import pt as opt
from probabilisticteacher_amd import ops
from probabilisticteacher_amd.config import setup_cfg
from probabilisticteacher_amd.engine import PTrainer
from probabilisticteacher_amd.modeling import sampling
from tests.helpers import keyed_perm_source

dev = "cuda:0"
burn, batch = 50, 2
cfg = setup_cfg(os.path.join(ROOT, "configs/pt/final_s2c.yaml"), [
    "MODEL.DEVICE", dev, "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", burn, "SOLVER.AMP.ENABLED", True,
    "SOLVER.IMG_PER_BATCH_LABEL", batch, "SOLVER.IMG_PER_BATCH_UNLABEL", batch, "SOLVER.WARMUP_ITERS", 100,
    "SOLVER.BASE_LR", 0.004])
K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
ocfg = opt.Cfg(num_classes=K, anchor_generator=cfg.MODEL.ANCHOR_GENERATOR.NAME, burn_up_step=burn,
               tau=tuple(cfg.UNSUPNET.TAU), ema_keep_rate=cfg.UNSUPNET.EMA_KEEP_RATE, base_lr=cfg.SOLVER.BASE_LR,
               warmup_iters=cfg.SOLVER.WARMUP_ITERS)
params = opt.golden_params(ocfg, 101)
ratio_rng = random.Random(5)
ratios = []
tr = PTrainer(cfg, ratio_fn=lambda: ratios.pop(0))
sd, tsd = tr.model.state_dict(), tr.model_teacher.state_dict()
with torch.no_grad():
    for k, v in params.items():
        sd[k].copy_(v); tsd[k].copy_(v)
gen = torch.Generator().manual_seed(2024)
pool = [make_batch(gen, batch, 128, 160, K, dev) for _ in range(8)]
found = 0
for it in range(300):
    hb, _ = pool[it % len(pool)]
    b = it < burn
    r_lab = [ratio_rng.uniform(0.5, 1.0) for _ in range(2 * batch if b else batch)]
    r_unl = [] if b else [ratio_rng.uniform(0.5, 1.0) for _ in range(batch)]
    rs = (r_lab if b else r_unl + r_lab)
    snap = (tr.student.flat.clone(), tr.teacher.flat.clone(), tr.momentum_buf.clone(), tr.iter, tr._first_step)

    def step(mode):
        tr.operand_rounding = mode
        ratios[:] = list(rs)
        sampling.set_key_source(keyed_perm_source(opt.KeyedPerm(1000 + it, strict=False)))
        try:
            return tr.run_step(hb)
        finally:
            sampling.set_key_source(None)
    m = step("bf16")
    if not b and not np.isfinite(m.get("loss_cls_unsup", 0.0)):
        with torch.no_grad():
            tr.student.flat.copy_(snap[0]); tr.teacher.flat.copy_(snap[1]); tr.momentum_buf.copy_(snap[2])
        tr.iter, tr._first_step = snap[3], snap[4]
        me = step("bf16_emulate")
        print("it", it, "\n  native ", {k: round(v, 5) for k, v in m.items() if k != "data_time"},
              "\n  emulate", {k: round(v, 5) for k, v in me.items() if k != "data_time"}, flush=True)
        found += 1
        if found >= 4:
            break
print("done, found", found)
