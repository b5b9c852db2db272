#!/usr/bin/env python
"""Time ptmi_conv3x3_wino_wgrad (F(2x2,3x3) domain) against ptmi_conv3x3_wino4_wgrad (F(4x4,3x3) domain) on the trainable BASELINE
layer shapes (HIP events) and report the difference of the two results.
    python tools/exp/wino4w_bench.py [--n 48] [--layers conv3_2,conv4_2]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
LAYERS = {"conv3_1": (128, 256, 200, 333), "conv3_2": (256, 256, 200, 333), "conv4_1": (256, 512, 100, 166),
          "conv4_2": (512, 512, 100, 166), "conv5_1": (512, 512, 50, 83)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=48)
    ap.add_argument("--layers", default=",".join(LAYERS))
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--lib", default=None, help="a variant library (tools/exp/make_wino4_variant.py with W4FILE=wino4w): only its F(4x4,3x3) kernel is timed")
    a = ap.parse_args()
    from probabilisticteacher_amd import _lib, ops
    lib = _lib.load()
    if a.lib:
        import ctypes
        vlib = ctypes.CDLL(os.path.abspath(a.lib))
        vlib.ptmi_conv3x3_wino4_wgrad_ws_floats.restype = ctypes.c_int64
    for name in a.layers.split(","):
        cin, cout, h, w = LAYERS[name]
        gen = torch.Generator().manual_seed(1)
        x = torch.relu(torch.randn(a.n, cin, h, w, generator=gen)).to("cuda:0")
        dy = torch.randn(a.n, cout, h, w, generator=gen).to("cuda:0")
        fl = 2.0 * 9 * cin * cout * h * w * a.n
        line, outs = f"{name:8s} n={a.n:2d}", {}
        for kind in (("wino4",) if a.lib else ("wino", "wino4")):
            ws = torch.empty(getattr(lib, f"ptmi_conv3x3_{kind}_wgrad_ws_floats")(a.n, cin, cout, h, w), device="cuda:0")
            dw, db = torch.empty(cout, cin, 3, 3, device="cuda:0"), torch.empty(cout, device="cuda:0")

            def f():
                if a.lib:
                    vp = ctypes.c_void_p
                    rc = vlib.ptmi_conv3x3_wino4_wgrad(vp(x.data_ptr()), vp(dy.data_ptr()), vp(dw.data_ptr()), vp(db.data_ptr()), vp(ws.data_ptr()),
                                                       a.n, cin, cout, h, w, 0, vp(torch.cuda.current_stream().cuda_stream))
                    assert rc == 0
                    return
                _lib.call(f"ptmi_conv3x3_{kind}_wgrad", ops._ptr(x), ops._ptr(dy), ops._ptr(dw), ops._ptr(db), ops._ptr(ws), a.n, cin, cout,
                          h, w, 0, ops._stream())
            f()
            torch.cuda.synchronize()
            ms = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    f()
                e1.record()
                torch.cuda.synchronize()
                ms = min(ms, e0.elapsed_time(e1) / a.iters)
            outs[kind] = (dw.clone(), db.clone(), ms)
            line += f"  {kind}: {ms:7.3f} ms {fl / ms / 1e9:6.1f} TF/s direct-eq"
        if a.lib:
            print(line, flush=True)
            continue
        sc = outs["wino"][0].abs().max().item()
        line += (f"  speed-up {outs['wino'][2] / outs['wino4'][2]:.3f}  max|dW diff| / max|dW| {(outs['wino'][0] - outs['wino4'][0]).abs().max().item() / sc:.2e}"
                 f"  db diff {(outs['wino'][1] - outs['wino4'][1]).abs().max().item() / outs['wino'][1].abs().max().item():.2e}")
        print(line, flush=True)


if __name__ == "__main__":
    main()
