"""Liveness probe for a candidate loss-curve workload (HIP side only): per trajectory the fraction of mutual-learning iterations
in which every unsupervised term is finite and non-zero, and the mutual-learning means.
    python tools/exp/curve_probe.py [--amp] --set burn=300 --set iters=500 seed [seed ...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import curve_common as cc
from tests.test_config4_gpu import _hip_trajectory
args = sys.argv[1:]
amp = False
st = dict(cc.SETTINGS)
seeds = []
while args:
    a = args.pop(0)
    if a == "--amp":
        amp = True
    elif a == "--set":
        k, v = args.pop(0).split("=")
        st[k] = type(st[k])(v)
    else:
        seeds.append(int(a))
pool_raw, sched = cc.make_pool(st, 1), cc.ratio_schedule(st)
burn, n = st["burn"], st["iters"]
keys = [k + s for s in ("_sup", "_unsup") for k in cc.LOSS_KEYS]
print("settings", st, "amp", amp, flush=True)
for seed in seeds:
    h = _hip_trajectory(st, seed, pool_raw, sched, amp=amp)
    live = {k: round(float(np.mean(np.isfinite(h[k][burn:n]) & (np.abs(h[k][burn:n]) > 1e-12))), 2) for k in keys if k.endswith("_unsup")}
    print("amp" if amp else "fp32", seed, "live", live, {k: round(float(np.nanmean(h[k][burn:n])), 4) for k in keys}, flush=True)
