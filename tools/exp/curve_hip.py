"""HIP-side trajectories of the configs[4] loss-curve workload for arbitrary sampler seeds (diagnostic).

    python tools/exp/curve_hip.py [--mode fp32|bf16|bf16_emulate] seed [seed ...]

Per trajectory: the mutual-learning means of every loss term, the fraction of mutual-learning iterations in which each unsupervised
term is finite and non-zero ("live"), and the burn-in (second half) means."""
import sys, os, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import curve_common as cc
from tests.test_config4_gpu import _hip_trajectory
from tests.helpers import load
args = sys.argv[1:]
mode = "fp32"
if args and args[0] == "--mode":
    mode, args = args[1], args[2:]
st = dict(cc.SETTINGS)
z = load("loss_curve_s2c")
pool_raw, sched = cc.make_pool(st, 1), cc.ratio_schedule(st)
seeds = [int(a) for a in args]
burn, n = st["burn"], st["iters"]
keys = [k + s for s in ("_sup", "_unsup") for k in cc.LOSS_KEYS]
for seed in seeds:
    h = _hip_trajectory(st, seed, pool_raw, sched, amp=(mode != "fp32"), rounding=("bf16_emulate" if mode == "bf16_emulate" else None))
    live = {k: round(float(np.mean(np.isfinite(h[k][burn:n]) & (np.abs(h[k][burn:n]) > 1e-12))), 2) for k in keys if k.endswith("_unsup")}
    print(mode, seed, {k: round(float(np.nanmean(h[k][burn:n])), 4) for k in keys}, "live", live,
          "burn2", {k: round(float(np.nanmean(h[k][burn // 2:burn])), 4) for k in cc.LOSS_KEYS}, flush=True)
print("oracle", {k: [round(float(np.nanmean(np.asarray(z[f"{k}@{s}"])[burn:n])), 4) for s in cc.KEY_SEEDS] for k in keys})
