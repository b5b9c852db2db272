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
s = s.replace('''        issue(0);
        for (int chunk = 0; chunk < nChunks; ++chunk) {
            const int buf = chunk & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces of `buf` have landed
            if (edge && fix) {''', '''        issue(0);
        t0 = clock64();
        for (int chunk = 0; chunk < nChunks; ++chunk) {
            const int buf = chunk & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces of `buf` have landed
            TICK(t_wait)
            if (edge && fix) {''')
s = s.replace('''            __syncthreads();                                      // everyone's pieces landed; buf^1 is free
            if (chunk + 1 < nChunks) issue(buf ^ 1);
            if (MODE == 0) continue;''', '''            __syncthreads();                                      // everyone's pieces landed; buf^1 is free
            TICK(t_bar)
            if (chunk + 1 < nChunks) issue(buf ^ 1);
            TICK(t_issue)
            if (MODE == 0) continue;''')
s = s.replace('''                    if (MODE == 2) {
                        const float b1 = psl[2 * j * PLANE + ky * PWB + kx + 8];
                        acc01 = mfma32(a0, b1, acc01);
                        acc11 = mfma32(a1, b1, acc11);
                    }
                }
            }
        }
    };''', '''                    if (MODE == 2) {
                        const float b1 = psl[2 * j * PLANE + ky * PWB + kx + 8];
                        acc01 = mfma32(a0, b1, acc01);
                        acc11 = mfma32(a1, b1, acc11);
                    }
                }
            }
            TICK(t_comp)
        }
    };''')
s = s.replace('''    auto run = [&](auto mode_c) {''', '''    long long t_wait = 0, t_bar = 0, t_issue = 0, t_comp = 0, t0 = 0, t1;
    const long long w_begin = wall_clock64();
#define TICK(acc_) t1 = clock64(); acc_ += t1 - t0; t0 = t1;
    auto run = [&](auto mode_c) {''')
s = s.replace('''    if (mode == 0) return;
''', '''    if (lane == 0) {
        long long* sink = g_sink + ((size_t)blockIdx.x * NWAVE + wave) * 8;
        sink[0] = t_wait; sink[1] = t_bar; sink[2] = t_issue; sink[3] = t_comp; sink[4] = mode; sink[5] = w_begin; sink[6] = wall_clock64();
    }
    if (mode == 0) return;
''')
s = s.replace("""    constexpr int CK = 4;
    constexpr int NT = 64 * NWAVE;
    constexpr int WN = NWAVE / (BM / 64);            // waves along the pixel dimension: 2 -> 4 rows, 4 -> 8 rows""", """    const long long w_entry = wall_clock64();
    constexpr int CK = 4;
    constexpr int NT = 64 * NWAVE;
    constexpr int WN = NWAVE / (BM / 64);            // waves along the pixel dimension: 2 -> 4 rows, 4 -> 8 rows""")
s = s.replace("sink[5] = w_begin; sink[6] = wall_clock64();", "sink[5] = w_begin; sink[6] = wall_clock64(); sink[7] = w_entry;")
s = s.replace("""                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)pv[q], soff, 0);
            }
        }
    }
}
""", """                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)pv[q], soff, 0);
            }
        }
    }
    const long long w_issued = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) { g_sink2[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NWAVE + wave] = wall_clock64(); 
                     g_sink3[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NWAVE + wave] = w_issued; }
}
""")
s = s.replace("((size_t)blockIdx.x * NWAVE + wave) * 8", "(((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NWAVE + wave) * 8")
s = s.replace('#include "common.h"', f'#include "{ROOT}/probabilisticteacher_amd/csrc/common.h"\n__device__ long long g_sink[1 << 22];\n__device__ long long g_sink2[1 << 19];\n__device__ long long g_sink3[1 << 19];\n'
              'extern "C" int exp_copy_sink3(void* dst, long long n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sink3), n * 8); }\n'
              'extern "C" int exp_copy_sink2(void* dst, long long n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sink2), n * 8); }\n'
              'extern "C" int exp_copy_sink(void* dst, long long n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sink), n * 8); }')
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
import sys
for name, cin, cout, h, w in (("conv2_2", 128, 128, 400, 666), ("conv3_2", 256, 256, 200, 333), ("conv4_2", 512, 512, 100, 166),
                              ("conv5_1", 512, 512, 50, 83)):
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
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
    args = (vp(x.data_ptr()), vp(wp.data_ptr()), vp(b.data_ptr()), None, vp(y.data_ptr()), n, cin, cout, h, w, 1, st)
    for _ in range(2):
        assert lib.ptmi_conv3x3_fwd(*args) == 0
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    lib.ptmi_conv3x3_fwd(*args)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    sink = torch.zeros(blocks * nw * 8, dtype=torch.int64)
    assert lib.exp_copy_sink(vp(sink.data_ptr()), sink.numel()) == 0
    t = sink.view(blocks, nw, 8).double()
    fl = 2.0 * cin * cout * 9 * h * w * n
    nch = cin // 4
    slots = 256 * (2 if use8 else 3)
    t_first, t_last = float(t[:, :, 5].min()), float(t[:, :, 6].max())
    span_us = (t_last - t_first) / 100.0
    blk_dur = (t[:, :, 6].max(1).values - t[:, :, 7].min(1).values) / 100.0          # us per block, entry .. loop end
    setup_us = float((t[:, :, 5] - t[:, :, 7]).mean()) / 100.0
    # per-slot gap analysis: sort blocks by entry time; at any time count resident blocks (entry..loop end)
    ent = t[:, :, 7].min(1).values
    end = t[:, :, 6].max(1).values
    ev = torch.cat([torch.stack([ent, torch.ones_like(ent)], 1), torch.stack([end, -torch.ones_like(end)], 1)])
    ev = ev[ev[:, 0].argsort()]
    resident = ev[:, 1].cumsum(0)
    dt = ev[1:, 0] - ev[:-1, 0]
    avg_res = float((resident[:-1] * dt).sum() / dt.sum())
    s2 = torch.zeros(blocks * nw, dtype=torch.int64)
    assert lib.exp_copy_sink2(vp(s2.data_ptr()), s2.numel()) == 0
    s2 = s2.view(blocks, nw).double()
    act = t[:, :, 4] > 0
    s3 = torch.zeros(blocks * nw, dtype=torch.int64)
    assert lib.exp_copy_sink3(vp(s3.data_ptr()), s3.numel()) == 0
    s3 = s3.view(blocks, nw).double()
    print(f"   epilogue: loop end .. last store issued {float(((s3 - t[:, :, 6])[act]).mean()) / 100.0:.2f} us, .. stores drained {float(((s2 - t[:, :, 6])[act]).mean()) / 100.0:.2f} us")
    epi_us = float(((s2 - t[:, :, 6])[act]).mean()) / 100.0
    ext = torch.where(act, s2, t[:, :, 6]).max(1).values
    full = float((ext - ent).sum()) / 100.0
    print(f"   epilogue (loop end .. stores drained) {epi_us:.2f} us per active wave; entry..exit occupancy {full / (span_us * slots):.3f}")
    print(f"   setup before first DMA {setup_us:.2f} us; time-averaged resident blocks (entry..loop-end) {avg_res:.1f} of {slots}; max {int(resident.max())}")
    mode_blk = t[:, :, 4].max(1).values
    print(f"{name} n={n}: kernel {ms:.3f} ms = {fl / ms / 1e9:.1f} TF/s; blocks {blocks} over {slots} slots = {blocks / slots:.2f} rounds; "
          f"first-start..last-end {span_us / 1e3:.3f} ms; slot occupancy {float(blk_dur.sum()) / (span_us * slots):.3f}")
    for md in (2, 1, 0):
        sel = t[:, :, 4] == md
        if sel.any():
            m = t[sel][:, :4].mean(0) / nch
            print(f"   waves mode {md}: {int(sel.sum()):7d}  per chunk: waitcnt {m[0]:6.0f} barrier {m[1]:6.0f} dma-issue {m[2]:6.0f} mfma {m[3]:6.0f} total {float(m.sum()):6.0f}")
    for md in (2, 1, 0):
        sel = mode_blk == md
        if sel.any():
            print(f"   blocks whose busiest wave is mode {md}: {int(sel.sum()):6d}  mean duration {float(blk_dur[sel].mean()):8.1f} us")
