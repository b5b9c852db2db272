#!/usr/bin/env python
"""tools/gen_loss_curve_golden.py -- the ORACLE side of the bounded BASELINE configs[4] loss-curve test, run in the dev
container (CPU, ~12 min per 300-iteration trajectory on 8 threads; round 5's six 500-iteration trajectories: two processes of three
seeds each on 3 threads, ~85 min, then --merge): final_s2c.yaml (K = 1), burn-in then mutual learning on the workload of
tests/curve_common.py, once per sampler-key seed of curve_common.KEY_SEEDS (same data, same initial weights, different
random anchor / ROI subsets: the spread between these trajectories is the yardstick the test measures the HIP side with).
Commits only numbers: tests/golden/loss_curve_s2c.npz = per-iteration oracle losses of every trajectory (`<key>@<seed>`),
gradient norms, pseudo-label counts, the settings.

    python tools/gen_loss_curve_golden.py [--set key=value ...] [--seeds 1000,5000] [--dry]   (--dry: summary only)"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import d2, pt as opt  # noqa: E402
from tests import curve_common as cc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", action="append", default=[])
    ap.add_argument("--dry", action="store_true")
    ap.add_argument("--seeds", default="")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default="")
    ap.add_argument("--merge", nargs="*", default=[], help="merge the .npz files of several --seeds runs (run in parallel processes) "
                                                           "into tests/golden/loss_curve_s2c.npz, trajectories in curve_common.KEY_SEEDS order")
    a = ap.parse_args()
    if a.merge:
        parts = [np.load(f) for f in a.merge]
        for q in parts[1:]:
            assert (q["settings_keys"] == parts[0]["settings_keys"]).all() and (q["settings_vals"] == parts[0]["settings_vals"]).all()
        arrays = {k: q[k] for q in parts for k in q.files if "@" in k}
        have = sorted(int(s) for q in parts for s in q["seeds"])
        assert have == sorted(cc.KEY_SEEDS), (have, cc.KEY_SEEDS)
        out = a.out or os.path.join(ROOT, "tests", "golden", "loss_curve_s2c.npz")
        np.savez_compressed(out, settings_keys=parts[0]["settings_keys"], settings_vals=parts[0]["settings_vals"],
                            seeds=np.array(cc.KEY_SEEDS), **arrays)
        print("wrote", out, os.path.getsize(out), "bytes")
        return
    st = dict(cc.SETTINGS)
    for kv in a.set:
        k, v = kv.split("=")
        st[k] = type(st[k])(v)
    torch.set_num_threads(a.threads or max(2, min(os.cpu_count() or 2, 32)))
    from probabilisticteacher_amd.config import setup_cfg
    cfg = setup_cfg(os.path.join(ROOT, "configs/pt/final_s2c.yaml"), ["MODEL.VGG.PRETRAIN", ""])
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    ocfg = opt.Cfg(num_classes=K, anchor_generator=cfg.MODEL.ANCHOR_GENERATOR.NAME, burn_up_step=st["burn"],
                   tau=tuple(cfg.UNSUPNET.TAU), ema_keep_rate=cfg.UNSUPNET.EMA_KEEP_RATE, base_lr=st["base_lr"],
                   warmup_iters=st["warmup_iters"])
    seeds = [int(v) for v in a.seeds.split(",")] if a.seeds else list(cc.KEY_SEEDS)
    raw_pool = cc.make_pool(st, K)
    ratios = cc.ratio_schedule(st)
    keys = [k + s for s in ("", "_sup", "_unsup") for k in cc.LOSS_KEYS]
    out_arrays = {}
    for seed0 in seeds:
        params = opt.golden_params(ocfg, st["param_seed"])
        state = {"student": {k: v.clone() for k, v in params.items()}, "teacher": {k: v.clone() for k, v in params.items()},
                 "bufs": {}, "iter": 0}
        pool = []
        for streams in raw_pool:
            recs = []
            for s in streams:
                rs = []
                for r in s:
                    inst = opt.FreeInstances(tuple(r["image"].shape[-2:]))
                    inst.gt_boxes, inst.gt_classes = d2.Boxes(r["boxes"].clone()), r["classes"].clone()
                    rs.append({"image": r["image"], "height": r["image"].shape[-2], "width": r["image"].shape[-1], "instances": inst})
                recs.append(rs)
            pool.append(tuple(recs))
        curve = {k: np.full(st["iters"], np.nan, np.float64) for k in keys + ["grad_norm", "n_pseudo"]}
        t0 = time.time()
        for it in range(st["iters"]):
            kp = opt.KeyedPerm(seed0 + it, strict=False)
            r_lab, r_unl = ratios[it]
            m = opt.run_step(ocfg, state, pool[it % len(pool)], {"label": r_lab, "unlabel": r_unl}, perm_fn=kp)
            for k in keys + ["grad_norm"]:
                if k in m:
                    curve[k][it] = m[k]
            if it >= st["burn"]:
                curve["n_pseudo"][it] = float(sum(len(p) for p in state["last_pseudo"]))
            if it % 20 == 0 or it == st["iters"] - 1:
                print(f"seed {seed0} it {it:4d} {time.time() - t0:6.0f}s", {k: round(v, 4) for k, v in m.items()}, flush=True)
        ml = slice(st["burn"], st["iters"])
        summary = {k: {"finite_nonzero_frac": float(np.mean(np.isfinite(curve[k][ml]) & (np.abs(curve[k][ml]) > 1e-12))),
                       "mean_finite": float(np.nanmean(curve[k][ml])) if np.isfinite(curve[k][ml]).any() else None}
                   for k in keys if k.endswith("_unsup")}
        print(f"seed {seed0}: mutual-learning iterations, unsupervised terms:", summary)
        for k, v in curve.items():
            out_arrays[f"{k}@{seed0}"] = v
    if not a.dry:
        out = a.out or os.path.join(ROOT, "tests", "golden", "loss_curve_s2c.npz")
        np.savez_compressed(out, settings_keys=np.array(sorted(st)), settings_vals=np.array([float(st[k]) for k in sorted(st)]),
                            seeds=np.array(seeds), **out_arrays)
        print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
