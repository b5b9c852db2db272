"""Run the configs[3] per-GPU-share step test's harness and dump the first image whose HIP proposal stage differs from the oracle's
stage on the same inputs (gpurun_out/proposal_mismatch.pt) for offline analysis."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytest
from tests import test_baseline_size_gpu as T


class MP:
    def __init__(self): self.undo = []
    def setattr(self, obj, name, val):
        self.undo.append((obj, name, getattr(obj, name))); setattr(obj, name, val)
    def context(self): return self


mp = MP()
n_img, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 83
res = T.mutual_learning_step_vs_oracle(mp, "configs/pt/final_c2f.yaml", 800, 1333, n_img=n_img, seed=seed)
log = res["log"]
for k, ((dec, lg, sg, size, pre, post, training, (hb, hs)), _) in enumerate(zip(log.hip_in, log.ref_in)):
    r = log._f_ref(log.ocfg, dec.unsqueeze(0), lg.unsqueeze(0), [size], sg.unsqueeze(0), pre, post, training)[0]
    rb = r.proposal_boxes.tensor
    same = len(hb) == len(rb) and torch.equal(hb, rb)
    print("image", k, "hip", len(hb), "ref", len(rb), "same", same, flush=True)
    if not same:
        os.makedirs("gpurun_out", exist_ok=True)
        torch.save({"dec": dec, "lg": lg, "sg": sg, "size": size, "pre": pre, "post": post, "training": training, "hb": hb, "hs": hs,
                    "rb": rb, "rs": r.objectness_logits}, "gpurun_out/proposal_mismatch.pt")
        print("dumped image", k)
        break
