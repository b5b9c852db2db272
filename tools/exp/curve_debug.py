"""dump the HIP side of the configs[4] loss curve (tests/test_config4_gpu.py) to gpurun_out/hip_curve.npz"""
import os, sys, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pt as opt
from tests import curve_common as cc
from tests.helpers import keyed_perm_source
from probabilisticteacher_amd.config import setup_cfg
from probabilisticteacher_amd.engine import PTrainer
from probabilisticteacher_amd.modeling import sampling
from probabilisticteacher_amd.structures import Boxes, FreeInstances
DEV = "cuda:0"
st = dict(cc.SETTINGS)
st["key_seed0"] = int(os.environ.get("KEY_SEED0", 1000))
OUT = os.environ.get("CURVE_OUT", "gpurun_out/hip_curve.npz")
cfg = setup_cfg("configs/pt/final_s2c.yaml", ["MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", st["burn"],
                "SOLVER.IMG_PER_BATCH_LABEL", st["batch"], "SOLVER.IMG_PER_BATCH_UNLABEL", st["batch"],
                "SOLVER.WARMUP_ITERS", st["warmup_iters"], "SOLVER.BASE_LR", st["base_lr"]])
K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
params = opt.golden_params(opt.Cfg(num_classes=K, anchor_generator=cfg.MODEL.ANCHOR_GENERATOR.NAME), st["param_seed"])
ratios = []
npseudo = []
class T(PTrainer):
    def process_pseudo_label(self, proposals, proposal_type, m=""):
        out, nn = super().process_pseudo_label(proposals, proposal_type, m)
        fg = sum(int((p.scores_logists.argmax(1) != K).sum()) for p in out)
        npseudo.append((sum(len(p) for p in out), fg))
        return out, nn
tr = T(cfg, ratio_fn=lambda: ratios.pop(0))
for model in (tr.model, tr.model_teacher):
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in params.items():
            sd[k].copy_(v)
pool = []
for streams in cc.make_pool(st, K):
    recs = []
    for s in streams:
        rs = []
        for r in s:
            inst = FreeInstances(tuple(r["image"].shape[-2:]))
            inst.gt_boxes, inst.gt_classes = Boxes(r["boxes"].to(DEV)), r["classes"].to(DEV)
            rs.append({"image": r["image"].to(DEV), "height": r["image"].shape[-2], "width": r["image"].shape[-1], "instances": inst})
        recs.append(rs)
    pool.append(tuple(recs))
sched = cc.ratio_schedule(st)
keys = [k + s for s in ("", "_sup", "_unsup") for k in cc.LOSS_KEYS]
hip = {k: np.full(st["iters"], np.nan) for k in keys}
for it in range(st["iters"]):
    r_lab, r_unl = sched[it]
    ratios[:] = r_lab if it < st["burn"] else r_unl + r_lab
    kp = opt.KeyedPerm(st["key_seed0"] + it, strict=False)
    sampling.set_key_source(keyed_perm_source(kp))
    try:
        m = tr.run_step(pool[it % len(pool)])
    finally:
        sampling.set_key_source(None)
    for k in keys:
        if k in m:
            hip[k][it] = m[k]
os.makedirs("gpurun_out", exist_ok=True)
np.savez(OUT, npseudo=np.array(npseudo), **hip)
print("done")
