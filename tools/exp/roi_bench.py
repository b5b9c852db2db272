#!/usr/bin/env python
"""tools/exp/roi_bench.py -- ROIAlign forward / backward at the step's launch shapes, ROI extents drawn like the bench's
(tools/exp/roi_stats.py: median 11 x 12.5 feature cells, 1 % wider than 55).   python tools/exp/roi_bench.py [--lib other.so]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from probabilisticteacher_amd import _lib, ops  # noqa: E402


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


def rois_like_the_step(n, per, g):
    # log-normal extents around 180 x 200 px, clipped to the image; centres uniform
    w = torch.exp(torch.randn(n * per, generator=g) * 0.65 + 5.2).clamp(24, 1333)
    h = torch.exp(torch.randn(n * per, generator=g) * 0.65 + 5.3).clamp(24, 800)
    cx, cy = torch.rand(n * per, generator=g) * 1333, torch.rand(n * per, generator=g) * 800
    x1, x2 = (cx - w / 2).clamp(0, 1333), (cx + w / 2).clamp(0, 1333)
    y1, y2 = (cy - h / 2).clamp(0, 800), (cy + h / 2).clamp(0, 800)
    img = torch.arange(n).repeat_interleave(per).float()
    return torch.stack([img, x1, y1, x2, y2], 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    for n, per, grad in ((32, 512, True), (16, 2000, False)):
        feat = torch.randn(n, 512, 50, 83, device=dev, requires_grad=grad)
        rois = rois_like_the_step(n, per, g).to(dev)
        wc = (rois[:, 3] - rois[:, 1]) / 16
        hc = (rois[:, 4] - rois[:, 2]) / 16
        offs = torch.arange(0, (n + 1) * per, per, dtype=torch.int32, device=dev)
        out = ops.roi_align(feat, rois, 7, 1 / 16, offs)
        ms = timeit(lambda: ops.roi_align(feat, rois, 7, 1 / 16, offs), a.iters)
        print(f"n={n} R={n * per} (median {float(wc.median()):.1f} x {float(hc.median()):.1f} cells): fwd {ms:7.3f} ms "
              f"({out.numel() * 4 / ms / 1e6:6.1f} GB/s of output)", end="")
        if grad:
            go = torch.randn_like(out)
            ms = timeit(lambda: torch.autograd.grad(out, feat, go, retain_graph=True), a.iters)
            print(f"   bwd {ms:7.3f} ms", end="")
        print(flush=True)
        del feat, out


if __name__ == "__main__":
    main()
