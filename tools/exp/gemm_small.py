#!/usr/bin/env python
"""tools/exp/gemm_small.py -- every fp32 GEMM shape of the step (box head + RPN 1x1), HIP-event timed: ms, TF/s, fraction of 157.3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from probabilisticteacher_amd import _lib, ops  # noqa: E402

if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
ONLY = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


dev = "cuda:0"
tot = 0.0
for name, k, nout in (("fc1", 25088, 1024), ("fc2", 1024, 1024), ("bbox_pred", 1024, 64), ("cls_score", 1024, 9)):
    if ONLY and name != ONLY:
        continue
    for r, grad in ((32000, False), (16384, True), (2400, True)):
        x = torch.randn(r, k, device=dev)
        w = torch.randn(nout, k, device=dev) * 0.01
        b = torch.zeros(nout, device=dev)
        dz = torch.randn(r, nout, device=dev)
        fl = 2.0 * r * k * nout
        rows = [("fwd", lambda: ops.gemm(x, w, r, nout, k, k, k, 0, 1, bias=b, bias_mode=2, relu=True))]
        if grad:
            rows += [("dX", lambda: ops.gemm(dz, w, r, k, nout, nout, k, 0, 0)), ("dW", lambda: ops.gemm(dz, x, nout, k, r, nout, k, 1, 0))]
        for what, fn in rows:
            ms = timeit(fn)
            tot += ms
            print(f"{name:10s} R={r:6d} {what:3s} {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF/s {fl / ms / 1e9 / 157.3:5.1%}")
        del x, dz
print(f"sum {tot:.2f} ms")
