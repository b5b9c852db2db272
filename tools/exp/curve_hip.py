"""HIP-side trajectories of the configs[4] loss-curve workload for arbitrary sampler seeds (diagnostic)."""
import sys, os, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import curve_common as cc
from tests.test_config4_gpu import _hip_trajectory
from tests.helpers import load
st = dict(cc.SETTINGS)
z = load("loss_curve_s2c")
pool_raw, sched = cc.make_pool(st, 1), cc.ratio_schedule(st)
seeds = [int(a) for a in sys.argv[1:]]
burn, n = st["burn"], st["iters"]
keys = [k + s for s in ("_sup", "_unsup") for k in cc.LOSS_KEYS]
for seed in seeds:
    h = _hip_trajectory(st, seed, pool_raw, sched)
    print(seed, {k: round(float(np.nanmean(h[k][burn:n])), 4) for k in keys}, "burn2", {k: round(float(np.nanmean(h[k][burn // 2:burn])), 4) for k in cc.LOSS_KEYS}, flush=True)
print("oracle", {k: [round(float(np.nanmean(np.asarray(z[f"{k}@{s}"])[burn:n])), 4) for s in cc.KEY_SEEDS] for k in keys})
