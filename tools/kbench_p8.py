#!/usr/bin/env python
"""tools/kbench_p8.py -- micro-benchmark of the bf16-storage (P8) conv kernels on the BASELINE shapes, HIP-event timed.

    python tools/kbench_p8.py [--n 48] [--which conv,dgrad,wgrad,pool] [--iters 5] [--layers conv3_2,conv4_2]

One line per layer: direct-convolution TFLOP/s and the fraction of the 2.5 PFLOP/s dense bf16 MFMA peak."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilisticteacher_amd import _lib, ops, p8  # noqa: E402

PEAK = 2500.0
LAYERS = [  # name, cin, cout, h, w
    ("conv1_1", 16, 64, 800, 1333), ("conv1_2", 64, 64, 800, 1333), ("conv2_1", 64, 128, 400, 666),
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
    ap.add_argument("--n", type=int, default=48)
    ap.add_argument("--which", default="conv,dgrad,wgrad")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--layers", default="")
    ap.add_argument("--lib", default="", help="another build of libptmi355.so (tools/exp/p8_variants.sh)")
    ap.add_argument("--zeros", action="store_true", help="zero-filled operands (DVFS check: the same kernel at lower power)")
    a = ap.parse_args()
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    dev = "cuda:0"
    which = a.which.split(",")
    sel = a.layers.split(",") if a.layers else None
    for name, cin, cout, h, w in LAYERS:
        if sel and name not in sel:
            continue
        n = a.n
        # random bf16 data (not zeros: DVFS), written straight into the P8 storage incl. the pads -- timing only
        x = (torch.randn((cin // 8, n * (h + 1) + 1, w + 1, 8), device=dev) * (0.0 if a.zeros else 0.5)).to(torch.bfloat16)
        wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        b = torch.zeros(cout, device=dev)
        fl = 2.0 * 9 * cin * cout * h * w * n
        if "conv" in which:
            wp = p8.pack_weights(wt, 0)
            ms = timeit(lambda: p8.conv3x3_raw(x, wp, b, None, n, cin, cout, h, w, 1), a.iters)
            print(f"{name:8s} fwd   n={n} {ms:9.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {fl / ms / 1e9 / PEAK:5.1%}", flush=True)
        if "dgrad" in which and cout % 16 == 0 and cin % 8 == 0 and cin >= 64:
            dy = (torch.randn((cout // 8, n * (h + 1) + 1, w + 1, 8), device=dev) * 0.5).to(torch.bfloat16)
            wpd = p8.pack_weights(wt, 1)
            ms = timeit(lambda: p8.conv3x3_raw(dy, wpd, None, x, n, cout, cin, h, w, 3), a.iters)
            print(f"{name:8s} dgrad n={n} {ms:9.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {fl / ms / 1e9 / PEAK:5.1%}", flush=True)
            del dy
        if "wgrad" in which and hasattr(p8, "wgrad") and cin >= 64:
            dy = (torch.randn((cout // 8, n * (h + 1) + 1, w + 1, 8), device=dev) * (0.0 if a.zeros else 0.5)).to(torch.bfloat16)
            try:
                ms = timeit(lambda: p8.wgrad(x, dy, n, cin, cout, h, w), a.iters)
                print(f"{name:8s} wgrad n={n} {ms:9.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {fl / ms / 1e9 / PEAK:5.1%}", flush=True)
            except _lib.PtmiError as e:          # blocks 1-2 at n = 48: beyond the 32-bit offsets (frozen in the step: never launched)
                print(f"{name:8s} wgrad n={n}   rejected: {str(e).split(': ', 1)[-1]}", flush=True)
            del dy
        if "pool" in which and name in ("conv1_2", "conv2_2", "conv3_2", "conv4_2"):
            xo = (torch.randn((cout // 8, n * (h + 1) + 1, w + 1, 8), device=dev) * 0.5).to(torch.bfloat16)
            ms = timeit(lambda: p8.maxpool_fwd(xo, n, cout, h, w), a.iters)
            gb = xo.numel() * 2 * 1.25 / 1e9
            print(f"{name:8s} pool  n={n} {ms:9.3f} ms  {gb / ms:7.2f} TB/s", flush=True)
            del xo
        del x
    if "gemm" in which:
        for r in (8192, 16384, 32000):
            k, nn = 25088, 1024
            xa = (torch.randn((k // 8, r, 8), device=dev) * 0.5).to(torch.bfloat16)
            wb = (torch.randn((k // 8, nn, 8), device=dev) * 0.05).to(torch.bfloat16)
            b1 = torch.zeros(nn, device=dev)
            fl = 2.0 * r * k * nn
            ms = timeit(lambda: p8.gemm_nt(xa, wb, r, nn, k, b1, True), a.iters)
            print(f"fc1 fwd  R={r:6d} {ms:9.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {fl / ms / 1e9 / PEAK:5.1%}", flush=True)
            dz = (torch.randn((nn // 8, r, 8), device=dev) * 0.5).to(torch.bfloat16)
            wt = (torch.randn((nn // 8, k, 8), device=dev) * 0.05).to(torch.bfloat16)
            ms = timeit(lambda: p8.gemm_nt(dz, wt, r, k, nn, swapped=True), a.iters)
            print(f"fc1 dx   R={r:6d} {ms:9.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {fl / ms / 1e9 / PEAK:5.1%}", flush=True)
            dzt = (torch.randn((r // 8, nn, 8), device=dev) * 0.5).to(torch.bfloat16)
            xt = (torch.randn((r // 8, k, 8), device=dev) * 0.5).to(torch.bfloat16)
            ms = timeit(lambda: p8.gemm_nt(dzt, xt, nn, k, r), a.iters)
            print(f"fc1 dw   R={r:6d} {ms:9.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {fl / ms / 1e9 / PEAK:5.1%}", flush=True)
            src = torch.randn(r, k, device=dev)
            ms = timeit(lambda: p8.pack_matrix(src, r, k, k, True), a.iters)
            ms2 = timeit(lambda: p8.pack_matrix(src, k, r, k, False), a.iters)
            print(f"fc1 pack R={r:6d} k-major {ms:7.3f} ms  row-major {ms2:7.3f} ms  ({src.numel() * 6 / 1e9:.2f} GB each)", flush=True)
            del xa, dz, dzt, xt, src


if __name__ == "__main__":
    main()
