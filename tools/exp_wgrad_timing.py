#!/usr/bin/env python
"""Experiment harness (GPU box): clock64()-instrumented copy of the buffer-DMA wgrad kernel; per wave and per
32-pixel stage, cycles spent in s_waitcnt / s_barrier / DMA issue / MFMA block.  Variants (argv[1], comma separated):
EXP_NO_MFMA drops the MFMA block (shows the cost of the DMA pipeline alone).  Not part of the product.

The predecessor of this kernel (global_load_lds, per-lane 4-B gathers) measured 90 / 3600 / 11100 / 9360 cycles per
stage (waitcnt / barrier / DMA issue / MFMA; ideal 18432 with four waves per SIMD) -- its DMA issue phase cost as much
wave time as the MFMA block, which is what the buffer-DMA rewrite removed."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, "probabilisticteacher_amd", "csrc", "conv.hip")).read()
old = '''        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wv < 33) {
            // right-edge tile'''
assert old in s
s = s.replace(old, '''        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TICK(t_wait)
        if (wv < 33) {
            // right-edge tile''')
old = '''        __syncthreads();
        if (tile + S < nTiles) issue(nn, nty, ntx, buf ^ 1);
        if (!EDGE)'''
assert old in s
s = s.replace(old, '''        __syncthreads();
        TICK(t_bar)
        if (tile + S < nTiles) issue(nn, nty, ntx, buf ^ 1);
        TICK(t_issue)
#ifndef EXP_NO_MFMA
        if (!EDGE)''')
old = '''                                acc, (wv + 1) >> 1);
        tx = ntx; ty = nty; n = nn;
    }'''
assert old in s
s = s.replace(old, '''                                acc, (wv + 1) >> 1);
#else
        acc[0][0] += lds[buf * WB_STAGE + a_off] + lds[buf * WB_STAGE + b_off];
#endif
        TICK(t_comp)
        tx = ntx; ty = nty; n = nn;
    }
    if (lane == 0) {
        long long* sink = g_sink + ((size_t)blockIdx.x * 8 + wave) * 6 + (EDGE ? 6 * 65536 : 0);
        sink[0] = t_wait; sink[1] = t_bar; sink[2] = t_issue; sink[3] = t_comp; sink[4] = it; sink[5] = 0;
    }''')
old = '''    int tile = s, it = 0;
    if (tile < nTiles) issue(n, ty, tx, 0);'''
assert old in s
s = s.replace(old, '''    int tile = s, it = 0;
    long long t_wait = 0, t_bar = 0, t_issue = 0, t_comp = 0, t0, t1;
#define TICK(acc_) t1 = clock64(); acc_ += t1 - t0; t0 = t1;
    if (tile < nTiles) issue(n, ty, tx, 0);
    t0 = clock64();''')
s = s.replace('#include "common.h"', f'#include "{ROOT}/probabilisticteacher_amd/csrc/common.h"\n__device__ long long g_sink[1 << 20];\n'
              'extern "C" int exp_copy_sink(void* dst, long long n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sink), n * 8); }', 1)
os.makedirs("/tmp/exp", exist_ok=True)
open("/tmp/exp/wgrad_timing.hip", "w").write(s)
abi = os.path.join(ROOT, "probabilisticteacher_amd", "csrc", "abi.cpp")
variant = sys.argv[1] if len(sys.argv) > 1 else ""
flags = [f"-D{v}" for v in variant.split(",") if v]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared"] + flags +
                      ["-x", "hip", "/tmp/exp/wgrad_timing.hip", abi, "-o", f"/tmp/exp/libwtiming{variant}.so"])
lib = ctypes.CDLL(f"/tmp/exp/libwtiming{variant}.so")
print("variant:", variant or "baseline")
vp = ctypes.c_void_p
lib.ptmi_conv3x3_wgrad_ws_floats.restype = ctypes.c_int64
dev = "cuda:0"
for name, cin, cout, h, w in (("conv3_2", 256, 256, 200, 333), ("conv4_2", 512, 512, 100, 166), ("conv5_1", 512, 512, 50, 83)):
    n = 16
    x = torch.randn(n, cin, h, w, device=dev)
    dy = torch.randn(n, cout, h, w, device=dev)
    dw = torch.empty(cout, cin, 3, 3, device=dev)
    ws = torch.empty(lib.ptmi_conv3x3_wgrad_ws_floats(n, cin, cout, h, w), device=dev)
    st = vp(torch.cuda.current_stream().cuda_stream)
    args = (vp(x.data_ptr()), vp(dy.data_ptr()), vp(dw.data_ptr()), None, vp(ws.data_ptr()), n, cin, cout, h, w, 0, st)
    for _ in range(2):
        assert lib.ptmi_conv3x3_wgrad(*args) == 0
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    lib.ptmi_conv3x3_wgrad(*args)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    sink = torch.zeros(1 << 20, dtype=torch.int64)
    assert lib.exp_copy_sink(vp(sink.data_ptr()), sink.numel()) == 0
    fl = 2.0 * cin * cout * 9 * h * w * n
    print(f"{name}: {ms:.3f} ms = {fl / ms / 1e9:.1f} TF/s (main + edge + reduce)")
    for tag, off in (("main", 0), ("edge", 6 * 65536)):
        t = sink[off:off + 6 * 65536].view(-1, 6).double()
        t = t[t[:, 4] > 0]
        if t.numel() == 0:
            continue
        m = t[:, :4].sum(0) / t[:, 4].sum()
        print(f"   {tag}: waves {t.shape[0]} stages/wave {t[:, 4].mean():.0f}; per stage per wave: waitcnt {m[0]:.0f} barrier {m[1]:.0f} "
              f"dma-issue {m[2]:.0f} mfma {m[3]:.0f} total {float(m.sum()):.0f} (main stage ideal: 72 MFMAs x 64 cyc x 4 waves/SIMD = 18432)")
