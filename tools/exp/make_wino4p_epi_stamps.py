#!/usr/bin/env python
"""A copy of csrc/wino4p.hip with s_memtime stamps of wave 0 of workgroup 0 INSIDE the epilogue (after the first send and after each
of the eight finalised channels: int64 slot 512 + 16 * tile + k of the buffer passed as mask_ref), to be built by
    W4FILE=wino4p W4SRC=/tmp/wino4p_epi.hip python tools/exp/make_wino4_variant.py -fno-slp-vectorize epistamp stamp
and read by  tools/exp/wino4_bench.py --only4 --p --stamps --lib tools/exp/_bin/libptmi355_wino4p_epistamp.so"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "probabilisticteacher_amd", "csrc", "wino4p.hip")).read()


def st(k):
    return ('if (blockIdx.x == 0 && tid == 0 && x4_tile < 64) { unsigned long long t_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); '
            f'((unsigned long long*)mref)[512 + 16 * x4_tile + {k}] = t_; }}')


a = "        send(std::integral_constant<int, 0>{});\n"
assert src.count(a) == 1
src = src.replace(a, "        " + st(8) + "\n" + a + "        " + st(9) + "\n")
b = "                __builtin_amdgcn_sched_barrier(0);\n            });\n        };\n        if (epi == 4) {"
assert src.count(b) == 1
src = src.replace(b, "                " + st("decltype(c_c)::value") + "\n" + b)
open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/wino4p_epi.hip", "w").write(src)
