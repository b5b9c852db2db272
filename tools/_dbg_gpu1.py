import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from oracle import d2, pt as opt
from tests.helpers import load, records
from probabilisticteacher_amd import ops, modeling
from probabilisticteacher_amd.config import setup_cfg
from probabilisticteacher_amd.modeling import sampling
from probabilisticteacher_amd.structures import FreeInstances
DEV="cuda:0"
z = load("model_default_anchor")
print("ref perm log", list(z["sup_perm_log"]))
cfg = setup_cfg("configs/pt/final_c2f.yaml", ["MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", ""])
ocfg = opt.Cfg(num_classes=8)
model = modeling.build_model(cfg).train()
params = opt.golden_params(ocfg, int(z["seed"]))
sd = model.state_dict()
with torch.no_grad():
    for k,v in params.items(): sd[k].copy_(v)
recs = records(z, "sup", 2, make_instances=FreeInstances)
orecs = records(z, "sup", 2)
perm = opt.SeededPerm(77); sampling.set_perm_fn(perm)
images = model.preprocess_image(recs)
oimg = opt.preprocess_image(ocfg, orecs)
print("preproc equal", torch.equal(images.tensor.cpu(), oimg.tensor))
feat = model.backbone(images.tensor)["vgg_block5"]
ofeat = opt.vgg_forward(params, oimg.tensor)
print("feat maxabs", float(ofeat.abs().max()), "maxdiff", float((feat.cpu()-ofeat).abs().max()))
anch = model.proposal_generator.anchor_generator([feat])[0].tensor
oanch = opt.make_anchors(ocfg, params, ofeat.shape[-2:], False)
print("anchors equal", torch.equal(anch.cpu(), oanch))
for i in range(2):
    gt = recs[i]["instances"].gt_boxes.tensor.to(DEV)
    midx, lab, iou = ops.iou_match(gt, anch, [0.3,0.7],[0,-1,1], True)
    m = d2.pairwise_iou(d2.Boxes(gt.cpu()), d2.Boxes(oanch))
    ridx, rlab = d2.Matcher([0.3,0.7],[0,-1,1],True)(m)
    print("img", i, "labels equal", torch.equal(lab.cpu(), rlab), "pos", int((lab==1).sum()), int((rlab==1).sum()), "neg", int((lab==0).sum()), int((rlab==0).sum()))
losses,_,_,_ = model(recs, branch="supervised")
print("hip perm log", perm.log)
print({k: float(v) for k,v in losses.items()})
print({k: float(z["sup_"+k]) for k in losses})
