#!/usr/bin/env python
"""Time ptmi_conv3x3_wino_wgrad of one or more builds of libptmi355.so on BASELINE layer shapes (HIP events).
    python tools/exp/wgrad_bench.py [--n 16] [--layers conv3_2,conv4_2] lib1.so lib2.so ..."""
import argparse
import ctypes
import os
import sys

import torch

LAYERS = {"conv1_2": (64, 64, 800, 1333), "conv2_2": (128, 128, 400, 666),
          "conv3_1": (128, 256, 200, 333), "conv3_2": (256, 256, 200, 333), "conv4_1": (256, 512, 100, 166),
          "conv4_2": (512, 512, 100, 166), "conv5_1": (512, 512, 50, 83)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--layers", default="conv3_2")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    vp = ctypes.c_void_p
    for name in a.layers.split(","):
        cin, cout, h, w = LAYERS[name]
        x = torch.randn(a.n, cin, h, w, device="cuda:0")
        dy = torch.randn(a.n, cout, h, w, device="cuda:0")
        dw = torch.empty(cout, cin, 3, 3, device="cuda:0")
        db = torch.empty(cout, device="cuda:0")
        fl = 2.0 * 9 * cin * cout * h * w * a.n
        for path in a.libs:
            lib = ctypes.CDLL(os.path.abspath(path))
            lib.ptmi_conv3x3_wino_wgrad_ws_floats.restype = ctypes.c_int64
            ws = torch.empty(lib.ptmi_conv3x3_wino_wgrad_ws_floats(a.n, cin, cout, h, w), device="cuda:0")
            st = vp(torch.cuda.current_stream().cuda_stream)

            def f():
                rc = lib.ptmi_conv3x3_wino_wgrad(vp(x.data_ptr()), vp(dy.data_ptr()), vp(dw.data_ptr()), vp(db.data_ptr()),
                                                 vp(ws.data_ptr()), a.n, cin, cout, h, w, 0, st)
                assert rc == 0
            f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print(f"{name:8s} n={a.n} {os.path.basename(path):28s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s effective", flush=True)


if __name__ == "__main__":
    main()
