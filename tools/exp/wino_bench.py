#!/usr/bin/env python
"""Time the Winograd kernel of one or more builds of libptmi355.so on BASELINE layer shapes (HIP events).
    python tools/exp/wino_bench.py [--n 16] [--layers conv3_2,conv4_2] lib1.so lib2.so ..."""
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
    ap.add_argument("--layers", default="conv3_2")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--trace", action="store_true", help="libs built with -DWINO_TRACE: print the k-step timestamps of one wave")
    ap.add_argument("--slots", action="store_true", help="libs built from wino_v5s.hip: slot-level stamps of KS1 / KS3")
    ap.add_argument("--coarse", action="store_true", help="libs built from wino_v5c.hip: entry / loop / epilogue stamps")
    ap.add_argument("--stream", action="store_true", help="libs built with -DWINO_TRACE from the streaming kernel: per-tile loop / epilogue stamps")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    vp = ctypes.c_void_p
    for name in a.layers.split(","):
        cin, cout, h, w = LAYERS[name]
        x = torch.randn(a.n, cin, h, w, device="cuda:0")
        wt = torch.randn(cout, cin, 3, 3, device="cuda:0") * 0.05
        b = torch.zeros(cout, device="cuda:0")
        y = torch.empty(a.n, cout, h, w, device="cuda:0")
        fl = 2.0 * 9 * cin * cout * h * w * a.n
        for path in a.libs:
            lib = ctypes.CDLL(os.path.abspath(path))
            lib.ptmi_conv3x3_wino_packed_floats.restype = ctypes.c_int64
            wp = torch.empty(lib.ptmi_conv3x3_wino_packed_floats(cin, cout), device="cuda:0")
            st = vp(torch.cuda.current_stream().cuda_stream)
            lib.ptmi_conv3x3_wino_pack_weights(vp(wt.data_ptr()), vp(wp.data_ptr()), cout, cin, 0, st)

            tr = torch.zeros(4096, dtype=torch.int64, device="cuda:0") if (a.trace or a.slots or a.coarse or a.stream) else None

            def f():
                rc = lib.ptmi_conv3x3_wino_fwd(vp(x.data_ptr()), vp(wp.data_ptr()), vp(b.data_ptr()),
                                               vp(tr.data_ptr()) if (a.trace or a.slots or a.coarse or a.stream) else None, vp(y.data_ptr()),
                                               a.n, cin, cout, h, w, 1, st)
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
            if a.stream:
                t = tr.cpu().tolist()
                for w in range(4):
                    v = t[w * 8: w * 8 + 4]
                    print(f"wave {w}: tile main loop {v[0]-v[3]}  epilogue issue {v[1]-v[0]}  store drain {v[2]-v[1]}")
            if a.coarse:
                t = tr.cpu().tolist()
                for g in range(2):
                    for w in range(4):
                        v = t[g * 32 + w * 8: g * 32 + w * 8 + 5]
                        print(f"wg {g} wave {w}: entry->loop {v[1]-v[0]}  loop {v[2]-v[1]}  epilogue issue {v[3]-v[2]}  store drain {v[4]-v[3]}  total {v[4]-v[0]}  entry abs {v[0]}")
            if a.slots:
                t = tr.cpu().tolist()
                for w in range(4):
                    for k, nm in ((0, "KS1"), (1, "KS3")):
                        v = t[w * 16 + k * 7: w * 16 + k * 7 + 7]
                        print(f"wave {w} {nm}: start->vmwait {v[1]-v[0] if v[1] else None} vmwait->after-barrier {v[2]-v[1] if v[2] else None} "
                              f"start->slot4 {v[3]-v[0]} slot4->8 {v[4]-v[3]} slot8->12 {v[5]-v[4]} slot12->15 {v[6]-v[5]}  abs start {v[0] % 100000}")
            if a.slots:
                for w in range(4):
                    tk0, t15p, tk015 = t[64 + w * 4: 64 + w * 4 + 3]
                    print(f"wave {w}: prev KS3 slot15 -> KS0 start {tk0 - t15p}; KS0 start -> KS0 slot15 {tk015 - tk0}; KS0 slot15 -> KS1 start {t[w * 16] - tk015}; "
                          f"KS1 slot15 -> (KS2) -> KS3 start {t[w * 16 + 7] - t[w * 16 + 6]}")
            if a.trace:
                t = tr.cpu().tolist()
                k = int(t[4095])
                d = [t[i + 1] - t[i] for i in range(0, k)]
                print("trace: prologue", d[0], "k-steps", d[1:k - 0][:40], "...", "mean k-step", sum(d[1:k]) / max(k - 1, 1),
                      "by KS:", [sum(d[1 + j:k:4]) / max(len(d[1 + j:k:4]), 1) for j in range(4)])
            print(f"{name} n={a.n} {os.path.basename(path):32s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s eff  "
                  f"{fl / 2.25 / ms / 1e9 / 157.3:6.1%} of MFMA peak (issued)")


if __name__ == "__main__":
    main()
