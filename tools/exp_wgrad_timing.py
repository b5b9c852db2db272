#!/usr/bin/env python
"""Experiment harness (GPU box): clock64()-instrumented copy of the LDS-DMA wgrad kernel; per wave and per 32-pixel
stage, cycles spent in s_waitcnt / s_barrier / DMA issue / MFMA block.  Not part of the product."""
import ctypes
import os
import subprocess

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, "probabilisticteacher_amd", "csrc", "conv.hip")).read()
s = s.replace('''    int tile = s, it = 0;
    if (tile < nTiles) issue(tile, 0);
    for (; tile < nTiles; tile += S, ++it) {
        const int buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile + S < nTiles) issue(tile + S, buf ^ 1);
        const float* al = lds + buf * WD_STAGE + a_off;
        const float* bl = lds + buf * WD_STAGE + b_off;
        if (tg == 0) wgrad_stage<0, 5>(al, bl, acc);
        else wgrad_stage<5, 4>(al, bl, acc);
    }''', '''    int tile = s, it = 0;
    long long t_wait = 0, t_bar = 0, t_issue = 0, t_comp = 0, t0, t1;
    const long long t_begin = clock64();
#define TICK(acc_) t1 = clock64(); acc_ += t1 - t0; t0 = t1;
    if (tile < nTiles) issue(tile, 0);
    t0 = clock64();
    for (; tile < nTiles; tile += S, ++it) {
        const int buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TICK(t_wait)
        __syncthreads();
        TICK(t_bar)
        if (tile + S < nTiles) issue(tile + S, buf ^ 1);
        TICK(t_issue)
        const float* al = lds + buf * WD_STAGE + a_off;
        const float* bl = lds + buf * WD_STAGE + b_off;
        if (tg == 0) wgrad_stage<0, 5>(al, bl, acc);
        else wgrad_stage<5, 4>(al, bl, acc);
        TICK(t_comp)
    }
    if (lane == 0) {
        long long* sink = g_sink + ((size_t)blockIdx.x * 8 + wave) * 6;
        sink[0] = t_wait; sink[1] = t_bar; sink[2] = t_issue; sink[3] = t_comp; sink[4] = it; sink[5] = t0 - t_begin;
    }''')
s = s.replace("dma4(d != -1 ? dyn + (d & 0xFFFFFF) : zero_page + lane, Dd + wave_base + i * 512);",
              "dma4(EXP_SRC(d != -1 ? dyn + (d & 0xFFFFFF) : zero_page + lane), Dd + wave_base + i * 512);")
s = s.replace("dma4(d != -1 ? xb + (d & 0xFFFFFF) : zero_page + lane, Xd + wave_base + i * 512);",
              "dma4(EXP_SRC(d != -1 ? xb + (d & 0xFFFFFF) : zero_page + lane), Xd + wave_base + i * 512);")
s = s.replace("""        if (tg == 0) wgrad_stage<0, 5>(al, bl, acc);
        else wgrad_stage<5, 4>(al, bl, acc);
        TICK(t_comp)""", """#ifndef EXP_NO_MFMA
        if (tg == 0) wgrad_stage<0, 5>(al, bl, acc);
        else wgrad_stage<5, 4>(al, bl, acc);
#else
        acc[0][0] += al[0] + bl[0];
#endif
        TICK(t_comp)""")
s = s.replace('#include "common.h"', f'#include "{ROOT}/probabilisticteacher_amd/csrc/common.h"\n__device__ long long g_sink[1 << 20];\n#ifdef EXP_ZERO_SRC\n#define EXP_SRC(p_) ((d != -1 || true) ? zero_page + lane : (p_))\n#else\n#define EXP_SRC(p_) (p_)\n#endif\n'
              'extern "C" int exp_copy_sink(void* dst, long long n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sink), n * 8); }')
os.makedirs("/tmp/exp", exist_ok=True)
open("/tmp/exp/wgrad_timing.hip", "w").write(s)
abi = os.path.join(ROOT, "probabilisticteacher_amd", "csrc", "abi.cpp")
import sys
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
    S = max(1, min(256, -(-1024 // ((cout // 128) * (cin // 32)))))
    blocks = (cout // 128) * (cin // 32) * S
    sink = torch.zeros(blocks * 8 * 6, dtype=torch.int64)
    assert lib.exp_copy_sink(vp(sink.data_ptr()), sink.numel()) == 0
    t = sink.view(-1, 6).double()
    stages = t[:, 4].mean()
    m = t[:, :4].sum(0) / t[:, 4].sum()
    fl = 2.0 * cin * cout * 9 * h * w * n
    print(f"{name}: {ms:.3f} ms = {fl / ms / 1e9:.1f} TF/s; blocks {blocks} S {S} stages/block {stages:.0f}; per stage per wave: "
          f"waitcnt {m[0]:.0f} barrier {m[1]:.0f} dma-issue {m[2]:.0f} mfma {m[3]:.0f} total {float(m.sum()):.0f} "
          f"(72 MFMA avg = 4608 cyc x 4 waves/SIMD = 18432); loop total {t[:, 5].mean() / 1e6:.3f} Mcyc")
