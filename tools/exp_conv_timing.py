#!/usr/bin/env python
"""Experiment harness (GPU box): builds a clock64()-instrumented copy of the LDS-DMA conv kernel and reports, per wave
and per K-chunk, the cycles spent in (a) s_waitcnt vmcnt(0), (b) s_barrier, (c) issuing the next chunk's DMA,
(d) the 72-MFMA block.  Not part of the product; used to decide what to optimise."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "probabilisticteacher_amd", "csrc", "conv.hip")
s = open(SRC).read()
s = s.replace('''    issue(0, 0);
    for (int chunk = 0; chunk < nChunks; ++chunk) {
        const int buf = chunk & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's DMA pieces of `buf` have landed
        __syncthreads();                                          // everyone's pieces landed; buf^1 is free
        if (chunk + 1 < nChunks) issue(chunk + 1, buf ^ 1);''', '''    long long t_wait = 0, t_bar = 0, t_issue = 0, t_comp = 0, t0, t1; const long long t_begin = clock64(); const long long w_begin = wall_clock64();
#define TICK(acc_) t1 = clock64(); acc_ += t1 - t0; t0 = t1;
    issue(0, 0);
    t0 = clock64();
    for (int chunk = 0; chunk < nChunks; ++chunk) {
        const int buf = chunk & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TICK(t_wait)
        __syncthreads();
        TICK(t_bar)
        if (chunk + 1 < nChunks) issue(chunk + 1, buf ^ 1);
        TICK(t_issue)''')
s = s.replace('''        if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
''', '''        if (PRIO) __builtin_amdgcn_s_setprio(0);
        TICK(t_comp)
    }
    if (lane == 0 && epi == 1 && mref) {
        long long* sink = (long long*)mref + ((size_t)blockIdx.x * NWAVE + wave) * 4;
        sink[0] = t_wait; sink[1] = t_bar; sink[2] = t_issue; sink[3] = t_comp;
        long long* sink2 = (long long*)mref + (size_t)gridDim.x * NWAVE * 4 + ((size_t)blockIdx.x * NWAVE + wave) * 2;
        sink2[0] = t0 - t_begin; sink2[1] = wall_clock64() - w_begin;
    }
''', 1)
s = s.replace('#include "common.h"', f'#include "{ROOT}/probabilisticteacher_amd/csrc/common.h"')
s = s.replace('PTMI_CHECK_ARG(epilogue != 3 || mask_ref, "conv3x3_fwd: mask_ref required for epilogue 3");', '')
os.makedirs("/tmp/exp", exist_ok=True)
open("/tmp/exp/conv_timing.hip", "w").write(s)
abi = os.path.join(ROOT, "probabilisticteacher_amd", "csrc", "abi.cpp")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
                       "-x", "hip", "/tmp/exp/conv_timing.hip", abi, "-o", "/tmp/exp/libtiming.so"])
lib = ctypes.CDLL("/tmp/exp/libtiming.so")
vp, i = ctypes.c_void_p, ctypes.c_int
lib.ptmi_conv3x3_packed_floats.restype = ctypes.c_int64
dev = "cuda:0"
for name, cin, cout, h, w in (("conv2_2", 128, 128, 400, 666), ("conv3_2", 256, 256, 200, 333), ("conv4_2", 512, 512, 100, 166),
                              ("conv5_1", 512, 512, 50, 83)):
    n = 16
    x = torch.randn(n, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    b = torch.zeros(cout, device=dev)
    wp = torch.empty(lib.ptmi_conv3x3_packed_floats(cin, cout), device=dev)
    st = vp(torch.cuda.current_stream().cuda_stream)
    lib.ptmi_conv3x3_pack_weights(vp(wt.data_ptr()), vp(wp.data_ptr()), cout, cin, 0, st)
    y = torch.empty(n, cout, h, w, device=dev)
    use8 = h >= 200
    th = 8 if use8 else 4
    nw = 8 if use8 else 4
    blocks = n * ((w + 31) // 32) * ((h + th - 1) // th) * (cout // 128)
    sink = torch.zeros(blocks * nw * 6, dtype=torch.int64, device=dev)
    for _ in range(2):
        rc = lib.ptmi_conv3x3_fwd(vp(x.data_ptr()), vp(wp.data_ptr()), vp(b.data_ptr()), vp(sink.data_ptr()), vp(y.data_ptr()),
                                  n, cin, cout, h, w, 1, st)
        assert rc == 0
    torch.cuda.synchronize()
    t = sink[:blocks * nw * 4].view(-1, 4).double()
    be = sink[blocks * nw * 4:].view(-1, 2)
    pro = be[:, 0].double() - t.sum(1)
    span = 0.0
    ghz = float((be[:, 0].double() / be[:, 1].double().clamp(min=1)).mean()) * 0.1
    loop_tot = float(t.sum(1).mean())
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    lib.ptmi_conv3x3_fwd(vp(x.data_ptr()), vp(wp.data_ptr()), vp(b.data_ptr()), vp(sink.data_ptr()), vp(y.data_ptr()),
                         n, cin, cout, h, w, 1, st)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    fl = 2.0 * cin * cout * 9 * h * w * n
    print(f"  kernel {ms:.3f} ms = {fl / ms / 1e9:.1f} TF/s; shader clock {ghz:.3f} GHz (clock64 / wall_clock64 @100MHz); "
          f"loop {loop_tot / 1e6:.3f} Mcyc/wave, prologue {float(pro.mean()):.0f} cyc/wave; blocks {blocks}")
    nch = cin // 4
    m = t.mean(0) / nch
    tot = float(m.sum())
    print(f"{name}: per chunk per wave cycles: waitcnt {m[0]:.0f}  barrier {m[1]:.0f}  dma-issue {m[2]:.0f}  mfma-block {m[3]:.0f}  "
          f"total {tot:.0f}  (72 MFMAs = 4608 busy cycles; {nw // 4 * 3 if not use8 else 4} waves share a SIMD)")
