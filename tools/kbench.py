#!/usr/bin/env python
"""tools/kbench.py -- per-kernel micro-benchmark on the BASELINE shapes (1333x800), HIP-event timed.

    python tools/kbench.py [--n 4] [--which conv,wgrad,gemm,roi] [--iters 3]

Prints one line per layer: algorithmic TFLOP/s and fraction of the 157.3 TFLOP/s fp32 MFMA peak."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilisticteacher_amd import _lib, ops  # noqa: E402

PEAK = 157.3
LAYERS = [  # name, cin, cout, h, w
    ("conv1_1", 3, 64, 800, 1333), ("conv1_2", 64, 64, 800, 1333), ("conv2_1", 64, 128, 400, 666),
    ("conv2_2", 128, 128, 400, 666), ("conv3_1", 128, 256, 200, 333), ("conv3_2", 256, 256, 200, 333),
    ("conv4_1", 256, 512, 100, 166), ("conv4_2", 512, 512, 100, 166), ("conv5_1", 512, 512, 50, 83),
]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--which", default="conv,wgrad,gemm,roi")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--layers", default="")
    ap.add_argument("--bf16", action="store_true", help="the FC GEMMs on ptmi_gemm_bf16 (fractions stay relative to the fp32 MFMA "
                                                        "peak); the bf16 3x3 convolutions are tools/kbench_p8.py's")
    ap.add_argument("--algo", default="auto", choices=["auto", "wino2", "direct"], help="fp32 forward / dgrad algorithm "
                    "(auto = fused Winograd F(4x4,3x3) / F(2x2,3x3) where they apply, wino2 = F(2x2,3x3) only; TF/s are ALGORITHMIC "
                    "direct-convolution FLOPs either way)")
    a = ap.parse_args()
    ops.set_conv_algo(a.algo)
    if a.bf16:
        ops.set_operand_rounding("bf16")
        a.which = ",".join(w for w in a.which.split(",") if w in ("gemm", "roi"))
    dev = "cuda:0"
    which = a.which.split(",")
    sel = a.layers.split(",") if a.layers else None
    for name, cin, cout, h, w in LAYERS:
        if sel and name not in sel:
            continue
        x = torch.randn(a.n, cin, h, w, device=dev)
        wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        b = torch.zeros(cout, device=dev)
        fl = 2.0 * 9 * cin * cout * h * w * a.n
        if "conv" in which:
            wp = ops.conv3x3_pack(wt, 0, 1)
            ms = timeit(lambda: ops.conv3x3_raw(x, wp, b, None, cout, 1), a.iters)
            print(f"{name:8s} fwd   n={a.n} {ms:9.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {fl / ms / 1e9 / PEAK:5.1%}")
        if "wgrad" in which and cin >= 64:
            dy = torch.randn(a.n, cout, h, w, device=dev)
            dw = torch.empty_like(wt)
            db = torch.empty(cout, device=dev)
            nws = _lib.load().ptmi_conv3x3_wgrad_ws_floats(a.n, cin, cout, h, w)
            ws = torch.empty(nws, device=dev)

            def f():
                _lib.call("ptmi_conv3x3_wgrad", ops._ptr(x), ops._ptr(dy), ops._ptr(dw), None, ops._ptr(ws), a.n, cin,
                          cout, h, w, 0, ops._stream())
            ms = timeit(f, a.iters)
            print(f"{name:8s} wgrad n={a.n} {ms:9.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {fl / ms / 1e9 / PEAK:5.1%}")
            if a.algo in ("auto", "wino2") and cout >= 64:
                ws2 = torch.empty(_lib.load().ptmi_conv3x3_wino_wgrad_ws_floats(a.n, cin, cout, h, w), device=dev)

                def f2():
                    _lib.call("ptmi_conv3x3_wino_wgrad", ops._ptr(x), ops._ptr(dy), ops._ptr(dw), None, ops._ptr(ws2), a.n,
                              cin, cout, h, w, 0, ops._stream())
                ms = timeit(f2, a.iters)
                print(f"{name:8s} wgrad n={a.n} {ms:9.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {fl / ms / 1e9 / PEAK:5.1%}  (winograd F(2x2,3x3) domain)")
                if _lib.load().ptmi_conv3x3_wino4_wgrad_fits(h, w):
                    ws3 = torch.empty(_lib.load().ptmi_conv3x3_wino4_wgrad_ws_floats(a.n, cin, cout, h, w), device=dev)

                    def f3():
                        _lib.call("ptmi_conv3x3_wino4_wgrad", ops._ptr(x), ops._ptr(dy), ops._ptr(dw), None, ops._ptr(ws3), a.n,
                                  cin, cout, h, w, 0, ops._stream())
                    ms = timeit(f3, a.iters)
                    print(f"{name:8s} wgrad n={a.n} {ms:9.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {fl / ms / 1e9 / PEAK:5.1%}  (winograd F(4x4,3x3) domain"
                          f"{'' if ops._wino4_wgrad_fill(h, w) >= ops._WINO4_WGRAD_MIN_FILL else '; NOT routed here: chunk fill < 0.9'})")
        del x
    if "gemm" in which:
        for r in (512 * a.n, 1024 * a.n, 2000 * a.n):
            xx = torch.randn(r, 25088, device=dev)
            w1 = torch.randn(1024, 25088, device=dev) * 0.01
            b1 = torch.zeros(1024, device=dev)
            fl = 2.0 * r * 25088 * 1024
            ms = timeit(lambda: ops.gemm(xx, w1, r, 1024, 25088, 25088, 25088, 0, 1, bias=b1, bias_mode=2, relu=True), a.iters)
            print(f"fc1 fwd  R={r:6d} {ms:9.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {fl / ms / 1e9 / PEAK:5.1%}")
            dz = torch.randn(r, 1024, device=dev)
            ms = timeit(lambda: ops.gemm(dz, w1, r, 25088, 1024, 1024, 25088, 0, 0), a.iters)
            print(f"fc1 dx   R={r:6d} {ms:9.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {fl / ms / 1e9 / PEAK:5.1%}")
            ms = timeit(lambda: ops.gemm(dz, xx, 1024, 25088, r, 1024, 25088, 1, 0), a.iters)
            print(f"fc1 dw   R={r:6d} {ms:9.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {fl / ms / 1e9 / PEAK:5.1%}")
            del xx, dz
    if "roi" in which:
        feat = torch.randn(a.n, 512, 50, 83, device=dev, requires_grad=True)
        g = torch.Generator().manual_seed(0)
        per = 512
        cx, cy = torch.rand(a.n * per, generator=g) * 1333, torch.rand(a.n * per, generator=g) * 800
        bw, bh = 32 + torch.rand(a.n * per, generator=g) * 300, 32 + torch.rand(a.n * per, generator=g) * 300
        img = torch.arange(a.n).repeat_interleave(per).float()
        rois = torch.stack([img, cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).to(dev)
        offs = torch.arange(0, (a.n + 1) * per, per, dtype=torch.int32, device=dev)
        out = ops.roi_align(feat, rois, 7, 1 / 16, offs)
        go = torch.randn_like(out)
        ms = timeit(lambda: ops.roi_align(feat, rois, 7, 1 / 16, offs), a.iters)
        print(f"roi_align fwd  R={a.n * per} {ms:8.3f} ms")
        ms = timeit(lambda: torch.autograd.grad(out, feat, go, retain_graph=True), a.iters)
        print(f"roi_align bwd (grouped, LDS) R={a.n * per} {ms:8.3f} ms")
        out2 = ops.roi_align(feat, rois, 7, 1 / 16, None)
        ms = timeit(lambda: torch.autograd.grad(out2, feat, go, retain_graph=True), a.iters)
        print(f"roi_align bwd (global atomics) R={a.n * per} {ms:8.3f} ms")


if __name__ == "__main__":
    main()
