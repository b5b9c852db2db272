#!/usr/bin/env python
"""Time ptmi_conv3x3_wino_fwd (F(2x2,3x3)) against ptmi_conv3x3_wino4_fwd (F(4x4,3x3)) on BASELINE layer shapes (HIP events),
and report the worst difference of the two outputs and (small n) the error of each against an fp64 convolution.
    python tools/exp/wino4_bench.py [--n 16] [--layers conv3_2,conv4_2] [--epi 1] [--lib path.so]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
LAYERS = {"conv1_2": (64, 64, 800, 1333), "conv2_1": (64, 128, 400, 666), "conv2_2": (128, 128, 400, 666),
          "conv3_1": (128, 256, 200, 333), "conv3_2": (256, 256, 200, 333), "conv4_1": (256, 512, 100, 166),
          "conv4_2": (512, 512, 100, 166), "conv5_1": (512, 512, 50, 83)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--layers", default=",".join(LAYERS))
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--epi", type=int, default=1)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--only4", action="store_true")
    ap.add_argument("--p", action="store_true", help="also time ptmi_conv3x3_wino4p_fwd (round 6: positions split over the wave pair)")
    ap.add_argument("--custom", action="append", default=[], help="name=cin,cout,h,w (repeatable)")
    ap.add_argument("--reps", type=int, default=3, help="timed repetitions; the minimum is reported")
    ap.add_argument("--sched", action="store_true", help="wino4: ALSO time ptmi_conv3x3_wino4_fwd_sched (the dynamic tile schedule of round 6)")
    ap.add_argument("--stamps", action="store_true", help="--lib built with the `stamp` part: per-tile phase stamps of wave 0 of workgroup 0")
    a = ap.parse_args()
    from probabilisticteacher_amd import _lib
    lib = ctypes.CDLL(os.path.abspath(a.lib)) if a.lib else _lib.load()
    vp = ctypes.c_void_p
    for c in a.custom:
        nm, v = c.split("=")
        LAYERS[nm] = tuple(int(t) for t in v.split(","))
    for name in ([c.split("=")[0] for c in a.custom] if a.custom else a.layers.split(",")):
        cin, cout, h, w = LAYERS[name]
        gen = torch.Generator().manual_seed(1)
        x = torch.relu(torch.randn(a.n, cin, h, w, generator=gen)).to("cuda:0")
        wt = (torch.randn(cout, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5).to("cuda:0")
        b = torch.zeros(cout, device="cuda:0")
        fl = 2.0 * 9 * cin * cout * h * w * a.n
        outs = {}
        line = f"{name:8s} n={a.n:2d}"
        for kind in ((("wino4",) if a.only4 else ("wino", "wino4")) + (("wino4p",) if a.p else ())):
            pf = getattr(lib, f"ptmi_conv3x3_{kind}_packed_floats")
            pf.restype = ctypes.c_int64
            wp = torch.empty(pf(cin, cout), device="cuda:0")
            st = vp(torch.cuda.current_stream().cuda_stream)
            getattr(lib, f"ptmi_conv3x3_{kind}_pack_weights")(vp(wt.data_ptr()), vp(wp.data_ptr()), cout, cin, 0, st)
            y = torch.empty(a.n, cout, h, w, device="cuda:0")
            fwd = getattr(lib, f"ptmi_conv3x3_{kind}_fwd")

            tr = torch.zeros(8 * 64 + 16 * 64, dtype=torch.int64, device="cuda:0")

            def f():
                rc = fwd(vp(x.data_ptr()), vp(wp.data_ptr()), vp(b.data_ptr()), vp(tr.data_ptr() if a.stamps else x.data_ptr()), vp(y.data_ptr()),
                         a.n, cin, cout, h, w, a.epi, st)
                assert rc == 0, _lib.load().ptmi_last_error()
            f()
            torch.cuda.synchronize()
            ms = 1e9
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    f()
                e1.record()
                torch.cuda.synchronize()
                ms = min(ms, e0.elapsed_time(e1) / a.iters)
            outs[kind] = (y, ms)
            if a.sched and kind == "wino4":
                sched = torch.zeros(16, dtype=torch.int32, device="cuda:0")
                y_static = y.clone()

                def fd():
                    rc = lib.ptmi_conv3x3_wino4_fwd_sched(vp(x.data_ptr()), vp(wp.data_ptr()), vp(b.data_ptr()), vp(x.data_ptr()), vp(y.data_ptr()),
                                                          a.n, cin, cout, h, w, a.epi, vp(sched.data_ptr()), st)
                    assert rc == 0, _lib.load().ptmi_last_error()
                fd()
                torch.cuda.synchronize()
                msd = 1e9
                for _ in range(a.reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.iters):
                        fd()
                    e1.record()
                    torch.cuda.synchronize()
                    msd = min(msd, e0.elapsed_time(e1) / a.iters)
                line += f"  wino4 dynamic: {msd:7.3f} ms (x{msd / ms:.3f} of static; equal {bool(torch.equal(y, y_static))})"
            if a.stamps and kind in ("wino4", "wino4p") and (kind == "wino4p" or not a.p):
                t = tr.cpu()[:512].view(64, 8).tolist()
                te = tr.cpu()[512:].view(64, 16).tolist()
                last = max(k for k in range(64) if t[k][0])
                if last > 2:
                    print(f"   shader clock over tiles 1 .. {last}: {(t[last][0] - t[1][0]) / (t[last][5] - t[1][5]) * 100:.0f} MHz "
                          f"(s_memtime ticks per 100 MHz s_memrealtime tick)")
                for k in range(1, 4):
                    if t[k][4] and t[k + 1][0]:
                        print(f"   tile {k}: zero-init {t[k][1]-t[k][0]}  chunk loop {t[k][2]-t[k][1]} ({(t[k][2]-t[k][1]) / (cin // 4):.0f} per chunk)  "
                              f"vmcnt(0) {t[k][3]-t[k][2]}  epilogue {t[k][4]-t[k][3]}  to next tile {t[k+1][0]-t[k][4]}  total {t[k+1][0]-t[k][0]}")
                        if te[k][8]:   # tools/exp/make_wino4p_epi_stamps.py: inside the epilogue
                            print(f"      epilogue: entry -> first send {te[k][8]-t[k][3]}  send {te[k][9]-te[k][8]}  channels "
                                  f"{[te[k][0]-te[k][9]] + [te[k][c]-te[k][c-1] for c in range(1, 8)]}  closing barrier {t[k][4]-te[k][7]}")
            line += f"  {kind}: {ms:7.3f} ms {fl / ms / 1e9:6.1f} TF/s direct-eq"
        if "wino4p" in outs and "wino4" in outs:
            d = (outs["wino4p"][0] - outs["wino4"][0]).abs().max().item()
            line += f"  wino4p/wino4 x{outs['wino4p'][1] / outs['wino4'][1]:.3f}  max |diff| {d:.2e}"
        if "wino" in outs and "wino4" in outs:
            d = (outs["wino"][0] - outs["wino4"][0]).abs().max().item()
            line += f"  speed-up {outs['wino'][1] / outs['wino4'][1]:.3f}  max |wino - wino4| {d:.2e} (max |y| {outs['wino'][0].abs().max().item():.2f})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
