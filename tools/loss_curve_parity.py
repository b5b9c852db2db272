#!/usr/bin/env python
"""Long-run loss-curve parity, HIP trainer vs CPU oracle (BASELINE.json configs[4] groundwork: "500-iter loss-curve parity
vs CPU ref").

Both sides train final_s2c.yaml (Sim10k->Cityscapes, K = 1) from the same seeded parameters on the same synthetic
records for N iterations -- burn-in, then EMA-teacher mutual learning -- each with ITS OWN teacher, proposals and
pseudo labels (no hand-over between the sides: end-to-end dynamics).  Shared: the data, the shrink ratios and the
sampler's random keys (the product draws them, the oracle replays them).  Two fp32 implementations of a chaotic
index-driven pipeline decorrelate sample by sample (one rank swap re-orders a proposal list), so per-iteration losses are
compared only over the first iterations; the long-run claim is statistical: smoothed curves agree.

    python tools/loss_curve_parity.py --iters 500 --out profiles/r02_loss_curve_s2c_fp32.json
    python tools/loss_curve_parity.py --iters 500 --bf16 --no-oracle ...   (SOLVER.AMP.ENABLED: native bf16 kernels, HIP only)
    python tools/loss_curve_parity.py --iters 500 --bf16 --emulate --no-oracle ...   (same numerics on the fp32 kernels)"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_batch(gen, n, h, w, K, dev):
    """n labelled (strong, weak) + n unlabelled (strong, weak) records; objects are bright rectangles on noise so that
    the detector has something to learn"""
    from oracle import d2, pt as opt
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    hip, ora = [[], [], [], []], [[], [], [], []]
    for stream in (0, 2):
        for _ in range(n):
            m = int(torch.randint(1, 4, (1,), generator=gen))
            xy = torch.rand(m, 2, generator=gen) * torch.tensor([w * 0.55, h * 0.55])
            wh = 24 + torch.rand(m, 2, generator=gen) * torch.tensor([w * 0.35, h * 0.35])
            boxes = torch.cat([xy, xy + wh], 1)
            cls = torch.randint(0, K, (m,), generator=gen)
            base = torch.randint(0, 96, (3, h, w), generator=gen, dtype=torch.uint8)
            for b in boxes.long().tolist():
                base[:, b[1]:b[3], b[0]:b[2]] += 120
            for view in (0, 1):                       # strong view = extra noise, weak view = the image
                img = base.clone()
                if view == 0:
                    img = (img.int() + torch.randint(-20, 21, img.shape, generator=gen)).clamp(0, 255).to(torch.uint8)
                a, o = FreeInstances((h, w)), opt.FreeInstances((h, w))
                a.gt_boxes, a.gt_classes = Boxes(boxes.clone().to(dev)), cls.clone().to(dev)
                o.gt_boxes, o.gt_classes = d2.Boxes(boxes.clone()), cls.clone()
                hip[stream + view].append({"image": img.to(dev), "height": h, "width": w, "instances": a})
                ora[stream + view].append({"image": img, "height": h, "width": w, "instances": o})
    return tuple(hip), tuple(ora)


def smooth(x, k):
    x = np.asarray(x, np.float64)
    c = np.cumsum(np.insert(x, 0, 0.0))
    return (c[k:] - c[:-k]) / k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--burn", type=int, default=50)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=160)
    ap.add_argument("--window", type=int, default=50)
    ap.add_argument("--bf16", action="store_true", help="HIP side: SOLVER.AMP.ENABLED (bf16-rounded conv / FC operands)")
    ap.add_argument("--emulate", action="store_true", help="with --bf16: round operands by tensor passes and run the fp32 "
                                                           "kernels (ops mode bf16_emulate) instead of the native bf16 kernels")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from oracle import pt as opt
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer
    from probabilisticteacher_amd.modeling import sampling
    from tests.helpers import keyed_perm_source
    dev = "cuda:0"
    torch.set_num_threads(max(2, min(os.cpu_count() or 2, 32)))
    cfg = setup_cfg(os.path.join(ROOT, "configs/pt/final_s2c.yaml"), [
        "MODEL.DEVICE", dev, "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", a.burn, "SOLVER.AMP.ENABLED", bool(a.bf16),
        "SOLVER.IMG_PER_BATCH_LABEL", a.batch, "SOLVER.IMG_PER_BATCH_UNLABEL", a.batch, "SOLVER.WARMUP_ITERS", 100,
        "SOLVER.BASE_LR", 0.004])
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    ocfg = opt.Cfg(num_classes=K, anchor_generator=cfg.MODEL.ANCHOR_GENERATOR.NAME, burn_up_step=a.burn,
                   tau=tuple(cfg.UNSUPNET.TAU), ema_keep_rate=cfg.UNSUPNET.EMA_KEEP_RATE, base_lr=cfg.SOLVER.BASE_LR,
                   warmup_iters=cfg.SOLVER.WARMUP_ITERS)
    params = opt.golden_params(ocfg, 101)
    ratio_rng = random.Random(5)
    ratios = []
    tr = PTrainer(cfg, ratio_fn=lambda: ratios.pop(0))
    if a.bf16 and a.emulate:
        tr.operand_rounding = "bf16_emulate"
    sd = tr.model.state_dict()
    tsd = tr.model_teacher.state_dict()
    with torch.no_grad():
        for k, v in params.items():
            sd[k].copy_(v)
            tsd[k].copy_(v)
    state = {"student": {k: v.clone() for k, v in params.items()}, "teacher": {k: v.clone() for k, v in params.items()},
             "bufs": {}, "iter": 0}
    gen = torch.Generator().manual_seed(2024)
    pool = [make_batch(gen, a.batch, a.height, a.width, K, dev) for _ in range(8)]
    keys = sorted(("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"))
    hip_curve, ora_curve, t_hip, t_ora = [], [], 0.0, 0.0
    for it in range(a.iters):
        hb, ob = pool[it % len(pool)]
        burn = it < a.burn
        n_lab = 2 * a.batch if burn else a.batch
        r_lab = [ratio_rng.uniform(0.5, 1.0) for _ in range(n_lab)]
        r_unl = [] if burn else [ratio_rng.uniform(0.5, 1.0) for _ in range(a.batch)]
        ratios[:] = (r_lab if burn else r_unl + r_lab)            # run_step resizes unlabel_q first (trainer.py:329-330)
        kp = opt.KeyedPerm(1000 + it, strict=False)
        sampling.set_key_source(keyed_perm_source(kp))
        t0 = time.perf_counter()
        try:
            m = tr.run_step(hb)
        finally:
            sampling.set_key_source(None)
        t_hip += time.perf_counter() - t0
        hip_curve.append(m)
        if not a.no_oracle:
            kp.start_replay()
            t0 = time.perf_counter()
            om = opt.run_step(ocfg, state, ob, {"label": r_lab, "unlabel": r_unl}, perm_fn=kp)
            t_ora += time.perf_counter() - t0
            ora_curve.append(om)
        if os.environ.get("PARITY_VERBOSE") and it >= a.burn - 1:
            print(it, {k: round(v, 5) for k, v in m.items() if k != "data_time"}, flush=True)
            if not a.no_oracle:
                print(it, "oracle", {k: round(v, 5) for k, v in om.items()}, flush=True)
        if it % 25 == 0 or it == a.iters - 1:
            print(f"it {it:4d} hip total {m['total_loss']:.5f}" + ("" if a.no_oracle else f"  oracle total {om['total_loss']:.5f}"),
                  flush=True)
    def finite_total(m):
        """sum of the finite loss terms: `loss_box_reg_unsup` is NaN by construction (mean over an empty tensor,
        fast_rcnn.py:262) whenever the teacher labels every matched ROI background -- reported, but no gradient flows"""
        return float(sum(v for k, v in m.items() if k[:4] == "loss" and np.isfinite(v)))

    def nan_keys(m):
        return sorted(k for k, v in m.items() if k[:4] == "loss" and not np.isfinite(v))

    mode = ", fp32"
    if a.bf16:
        mode = (", bf16 operands: rounding passes + fp32 kernels (bf16_emulate)" if a.emulate
                else ", bf16 operands: native v_mfma_f32_32x32x16_bf16 kernels")
    out = {"config": "final_s2c.yaml (K=1)" + mode,
           "iters": a.iters, "burn_up_step": a.burn, "batch": [a.batch, a.batch], "image": [a.height, a.width],
           "hip_finite_total_loss": [finite_total(m) for m in hip_curve], "hip_seconds": t_hip,
           "hip_grad_norm_finite": bool(all(np.isfinite(m["grad_norm"]) for m in hip_curve)),
           "hip_nan_terms": sorted({k for m in hip_curve for k in nan_keys(m)})}
    if a.iters >= 2 * a.window:
        hs = smooth(out["hip_finite_total_loss"], a.window)
        out["hip_smoothed"] = {"window": a.window, "first": float(hs[0]), "at_burn": float(hs[min(a.burn, len(hs) - 1)]),
                               "last": float(hs[-1])}
    if not a.no_oracle:
        h = np.array(out["hip_finite_total_loss"])
        o = np.array([finite_total(m) for m in ora_curve])
        out["oracle_finite_total_loss"] = o.tolist()
        out["oracle_seconds"] = t_ora
        out["oracle_grad_norm_finite"] = bool(all(np.isfinite(m["grad_norm"]) for m in ora_curve))
        out["oracle_nan_terms"] = sorted({k for m in ora_curve for k in nan_keys(m)})
        first = min(5, a.iters)
        out["first_iters_rel_diff"] = (np.abs(h[:first] - o[:first]) / np.abs(o[:first])).tolist()
        out["nan_term_rate"] = {"hip": float(np.mean([bool(nan_keys(m)) for m in hip_curve])),
                                "oracle": float(np.mean([bool(nan_keys(m)) for m in ora_curve]))}
        if a.iters >= 2 * a.window:
            hs, os_ = smooth(h, a.window), smooth(o, a.window)
            rel = np.abs(hs - os_) / np.abs(os_)
            out["smoothed_window"] = a.window
            out["smoothed_rel_diff_max"] = float(rel.max())
            out["smoothed_rel_diff_mean"] = float(rel.mean())
            out["final_window_mean"] = {"hip": float(hs[-1]), "oracle": float(os_[-1])}
            per_key = {}
            for k in sorted({k for m in hip_curve for k in m if k[:4] == "loss"}):
                hk = np.array([m.get(k, np.nan) for m in hip_curve], np.float64)
                ok_ = np.array([m.get(k, np.nan) for m in ora_curve], np.float64)
                msk = np.isfinite(hk) & np.isfinite(ok_)
                if msk.sum() >= a.window:
                    per_key[k] = {"hip_mean": float(hk[msk].mean()), "oracle_mean": float(ok_[msk].mean())}
            out["per_term_mean_over_common_finite_iters"] = per_key
    s = json.dumps(out)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(s + "\n")
    summary = {k: v for k, v in out.items() if not k.endswith("total_loss")}
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
