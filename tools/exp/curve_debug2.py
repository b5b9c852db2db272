"""After 100 HIP burn-in iterations: the HIP teacher vs the ORACLE teacher on the SAME weights and images (pseudo-label counts)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import d2, pt as opt
from tests import curve_common as cc
from tests.helpers import keyed_perm_source
from probabilisticteacher_amd.config import setup_cfg
from probabilisticteacher_amd.engine import PTrainer
from probabilisticteacher_amd.modeling import sampling
from probabilisticteacher_amd.structures import Boxes, FreeInstances
DEV = "cuda:0"
st = dict(cc.SETTINGS)
cfg = setup_cfg("configs/pt/final_s2c.yaml", ["MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", st["burn"],
                "SOLVER.IMG_PER_BATCH_LABEL", st["batch"], "SOLVER.IMG_PER_BATCH_UNLABEL", st["batch"],
                "SOLVER.WARMUP_ITERS", st["warmup_iters"], "SOLVER.BASE_LR", st["base_lr"]])
K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
ocfg = opt.Cfg(num_classes=K, anchor_generator=cfg.MODEL.ANCHOR_GENERATOR.NAME, burn_up_step=st["burn"], tau=tuple(cfg.UNSUPNET.TAU),
               ema_keep_rate=cfg.UNSUPNET.EMA_KEEP_RATE, base_lr=st["base_lr"], warmup_iters=st["warmup_iters"])
params = opt.golden_params(ocfg, st["param_seed"])
ratios = []
tr = PTrainer(cfg, ratio_fn=lambda: ratios.pop(0))
for model in (tr.model, tr.model_teacher):
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in params.items():
            sd[k].copy_(v)
raw = cc.make_pool(st, K)
def wrap(r, dev):
    inst = (FreeInstances if dev else opt.FreeInstances)(tuple(r["image"].shape[-2:]))
    if dev:
        inst.gt_boxes, inst.gt_classes = Boxes(r["boxes"].to(DEV)), r["classes"].to(DEV)
    else:
        inst.gt_boxes, inst.gt_classes = d2.Boxes(r["boxes"].clone()), r["classes"].clone()
    return {"image": r["image"].to(DEV) if dev else r["image"], "height": r["image"].shape[-2], "width": r["image"].shape[-1], "instances": inst}
pool = [tuple([wrap(r, True) for r in s] for s in streams) for streams in raw]
sched = cc.ratio_schedule(st)
for it in range(st["burn"]):
    r_lab, r_unl = sched[it]
    ratios[:] = r_lab
    kp = opt.KeyedPerm(st["key_seed0"] + it, strict=False)
    sampling.set_key_source(keyed_perm_source(kp))
    try:
        tr.run_step(pool[it % len(pool)])
    finally:
        sampling.set_key_source(None)
sd = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
tsd = tr.model_teacher.state_dict()
with torch.no_grad():
    for k in tsd:
        tsd[k].copy_(tr.model.state_dict()[k])
for b in range(4):
    weak_h = [{k: v for k, v in r.items() if k != "instances"} for r in pool[b][3]]
    weak_o = [{k: v for k, v in wrap(r, False).items() if k != "instances"} for r in raw[b][3]]
    with torch.no_grad():
        _, rp, roih, _ = tr.model_teacher(weak_h, branch="unsup_data_weak")
        _, orp, oroih, _ = opt.model_forward(ocfg, sd, weak_o, "unsup_data_weak", perm_fn=opt.SeededPerm(1))
    for i in range(len(roih)):
        hl, ol = roih[i].scores_logists.cpu(), oroih[i].scores_logists
        print(f"batch {b} img {i}: HIP {len(roih[i])} dets, fg-argmax {int((hl.argmax(1) != K).sum())}, max p_fg {float(torch.softmax(hl,1)[:,0].max()) if len(hl) else 0:.3f} | "
              f"oracle {len(oroih[i])} dets, fg-argmax {int((ol.argmax(1) != K).sum())}, max p_fg {float(torch.softmax(ol,1)[:,0].max()) if len(ol) else 0:.3f} | "
              f"proposals {len(rp[i])} vs {len(orp[i])}; HIP scores {[round(float(s),3) for s in roih[i].scores[:5]]} oracle {[round(float(s),3) for s in oroih[i].scores[:5]]}")
