"""Randomised soak of the F(4x4,3x3) kernels (the default forward / dgrad family -- csrc/wino4p.hip since round 6, both tile schedules -- and csrc/wino4w.hip) against the direct kernels of the same
library: random shapes (channel counts, map sizes incl. maps much smaller / larger than a workgroup tile, image counts that
make a persistent workgroup walk several tiles), every epilogue, forward and dgrad packs, the weight gradient; every launch is
issued TWICE and the two results must be bitwise equal (a race in the chunk stream would show as run-to-run differences long
before it shows against the 1e-4 bar).

    python tools/exp/wino4_fuzz.py [--seconds 240] [--seed 0]

Prints one line per 5000 cases and a summary; exit code 1 on the first failure (the failing shape is printed)."""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from probabilisticteacher_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def direct(fn):
    ops.set_conv_algo("direct")
    try:
        return fn()
    finally:
        ops.set_conv_algo("auto")


def fwd_case(rng, gen):
    big = rng.random() < 0.25
    cin = 8 * rng.randint(8, 64)
    cout = rng.choice([64, 128, 256, 512]) if rng.random() < 0.5 else rng.randint(1, 300)
    if big:
        n, h, w = rng.randint(2, 6), rng.randint(60, 200), rng.randint(100, 340)
        cin, cout = min(cin, 128), min(cout, 128)
    else:
        n, h, w = rng.randint(1, 4), rng.randint(1, 70), rng.randint(1, 180)
    mode = rng.randint(0, 1)
    epi = rng.choice([0, 1, 2, 4]) if mode == 0 else rng.choice([2, 3])
    if epi == 4 and (h < 2 or w < 2):
        epi = 1
    x = torch.randn(n, cin, h, w, generator=gen, device=DEV)
    wt = torch.randn((cout, cin, 3, 3) if mode == 0 else (cin, cout, 3, 3), generator=gen, device=DEV) * math.sqrt(2.0 / (9 * cin))
    bias = torch.randn(cout, generator=gen, device=DEV) * 0.1 if epi in (0, 1, 4) else None
    mask = torch.randn(n, cout, h, w, generator=gen, device=DEV) if epi == 3 else None
    what = f"fwd mode {mode} epi {epi} n {n} cin {cin} cout {cout} h {h} w {w}"
    assert ops._conv_kind(cin, cout, (h, w)) == "wino4", what

    sched = rng.choice(["static", "dynamic"])        # round 6: both tile schedules of the default family (ops._WINO4_FAMILY: wino4p)
    what += f" schedule {sched}"

    def run():
        ops.set_tile_schedule(sched)
        wp = ops.conv3x3_pack(wt, mode, epi, (h, w))
        if epi == 4:
            return ops.conv3x3_relu_pool_nograd(x, wt, bias)
        return ops.conv3x3_raw(x, wp, bias, mask, cout, epi)

    a, b = run(), run()
    if not torch.equal(a, b):
        return what + f": two launches differ (max {float((a - b).abs().max()):.3e})"
    ref = direct(run)
    ops.set_tile_schedule("static")
    e = rel_err(a, ref)
    if not (e <= 1e-4):
        return what + f": max err / max |ref| = {e:.3e}"
    return e


def wgrad_case(rng, gen):
    lib = _lib.load()
    cin = rng.choice([32, 64, 96, 128, 256, 40, 72])
    cout = rng.choice([64, 128, 192, 256, 80, 100])
    n, h, w = rng.randint(1, 5), rng.randint(1, 120), rng.randint(1, 340)
    if not lib.ptmi_conv3x3_wino4_wgrad_fits(h, w):
        return None
    x = torch.randn(n, cin, h, w, generator=gen, device=DEV)
    dz = torch.randn(n, cout, h, w, generator=gen, device=DEV)
    what = f"wgrad n {n} cin {cin} cout {cout} h {h} w {w}"

    def run(abi):
        dw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV)
        db = torch.full((cout,), float("nan"), device=DEV)
        ws = torch.empty(getattr(lib, abi + "_ws_floats")(n, cin, cout, h, w), device=DEV)
        _lib.call(abi, ops._ptr(x), ops._ptr(dz), ops._ptr(dw), ops._ptr(db), ops._ptr(ws), n, cin, cout, h, w, 0, ops._stream())
        return dw, db

    (a, ab), (b, bb) = run("ptmi_conv3x3_wino4_wgrad"), run("ptmi_conv3x3_wino4_wgrad")
    if not (torch.equal(a, b) and torch.equal(ab, bb)):
        return what + f": two launches differ (max {float((a - b).abs().max()):.3e})"
    ref, refb = run("ptmi_conv3x3_wgrad")
    e, eb = rel_err(a, ref), rel_err(ab, refb)
    if not (e <= 1e-4 and eb <= 1e-4):
        return what + f": dW err {e:.3e}, db err {eb:.3e} (relative to the largest entry)"
    return e


def main():
    import random
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    gen = torch.Generator(device=DEV).manual_seed(args.seed)
    t0 = time.time()
    n_f = n_w = 0
    worst_f = worst_w = 0.0
    shown = 0
    while time.time() - t0 < args.seconds:
        if rng.random() < 0.7:
            r = fwd_case(rng, gen)
            if isinstance(r, str):
                print("FAIL", r)
                return 1
            n_f += 1
            worst_f = max(worst_f, r)
        else:
            r = wgrad_case(rng, gen)
            if isinstance(r, str):
                print("FAIL", r)
                return 1
            if r is not None:
                n_w += 1
                worst_w = max(worst_w, r)
        if n_f + n_w >= shown + 5000:
            shown = n_f + n_w
            print(f"{time.time() - t0:6.0f} s  {n_f} fwd/dgrad cases (worst {worst_f:.2e})  {n_w} wgrad cases (worst {worst_w:.2e})", flush=True)
    print(f"OK: {n_f} forward / dgrad cases, worst max-err / max|ref| {worst_f:.2e}; {n_w} weight-gradient cases, worst {worst_w:.2e}; "
          f"every launch repeated, bitwise equal; seed {args.seed}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
