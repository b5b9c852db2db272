#!/usr/bin/env python
"""Write an INSTRUMENTED copy of csrc/p8.hip with elimination switches for timing experiments (-DP8X=<mask>; results WRONG by
design when non-zero; the product source carries none of this):
    1  forward / dgrad kernel: no DMA in the main loop          2  ... no operand LDS reads in the main loop
    4  ... no hand-over (vmcnt / barrier) in the main loop      8  ... no epilogue stores
   16  weight-gradient kernel: no DMA in the loop              32  ... no operand reads          64 ... no barrier

    python tools/exp/make_p8_variant.py tools/exp/_snap/p8_x.hip
    tools/exp/p8_variants.sh nodma=tools/exp/_snap/p8_x.hip:1 noread=tools/exp/_snap/p8_x.hip:2 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "probabilisticteacher_amd", "csrc", "p8.hip")

EDITS = [
    ("constexpr int P8T = 256;", "#ifndef P8X\n#define P8X 0\n#endif\nconstexpr int P8T = 256;"),
    ("            if constexpr (TAP < 6 && (J == 10 || J == 12 || J == 14))      // the next chunk into the other stage: everybody left it at the\n"
     "                dma(",
     "            if constexpr (TAP < 6 && (J == 10 || J == 12 || J == 14))      // the next chunk into the other stage: everybody left it at the\n"
     "                if (!(P8X & 1)) dma("),
    ("            if constexpr (J < MT) An[J] = read_a(src, NT_, J);\n            else if constexpr (J < MT + NTB) Bn[J - MT] = read_b(src, NT_, J - MT);",
     "            if (!(P8X & 2)) {\n            if constexpr (J < MT) An[J] = read_a(src, NT_, J);\n            else if constexpr (J < MT + NTB) Bn[J - MT] = read_b(src, NT_, J - MT);\n            }"),
    ("        if constexpr (TAP == 8) {\n            asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");",
     "        if constexpr (TAP == 8 && !(P8X & 4)) {\n            asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");"),
    ("                __builtin_amdgcn_raw_buffer_store_b128(w, ry, (int)svoff[j], 0, 0);",
     "                if (!(P8X & 8)) __builtin_amdgcn_raw_buffer_store_b128(w, ry, (int)svoff[j], 0, 0);\n                else asm volatile(\"\" :: \"v\"(w));"),
    ("                if (more && (s >> 1) == tg) {", "                if (!(P8X & 16) && more && (s >> 1) == tg) {"),
    ("                asm volatile(\"s_waitcnt lgkmcnt(0)\"\n                             : \"+v\"(ra[0])",
     "                if (P8X & 32) {\n#pragma unroll\n                    for (int i_ = 0; i_ < 4; ++i_) ra[i_] = (u32x2){0x3F803F80u, 0x3F803F80u};\n"
     "#pragma unroll\n                    for (int i_ = 0; i_ < 10; ++i_) rb[i_] = (u32x2){0x3F803F80u, 0x3F803F80u};\n                }\n"
     "                asm volatile(\"s_waitcnt lgkmcnt(0)\"\n                             : \"+v\"(ra[0])"),
    ("                asm volatile(\"ds_read_b64_tr_b16 %0, %1 offset:%2\" : \"=v\"(ra[0]) : \"v\"(sa + (unsigned)a_base), \"n\"(s * 256));",
     "                if (!(P8X & 32)) {\n                asm volatile(\"ds_read_b64_tr_b16 %0, %1 offset:%2\" : \"=v\"(ra[0]) : \"v\"(sa + (unsigned)a_base), \"n\"(s * 256));"),
    ("                    asm volatile(\"ds_read_b64_tr_b16 %0, %1 offset:%2\" : \"=v\"(rb[2 * t + 1]) : \"v\"(sa + (unsigned)b_base[t]), \"n\"(so + 64));\n                }",
     "                    asm volatile(\"ds_read_b64_tr_b16 %0, %1 offset:%2\" : \"=v\"(rb[2 * t + 1]) : \"v\"(sa + (unsigned)b_base[t]), \"n\"(so + 64));\n                }\n                }"),
]


def main():
    out = sys.argv[1]
    s = open(SRC).read()
    for old, new in EDITS:
        assert s.count(old) == 1, f"anchor not unique / missing in csrc/p8.hip ({s.count(old)}x): {old[:70]!r}"
        s = s.replace(old, new)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    open(out, "w").write(s)
    print(out)


if __name__ == "__main__":
    main()
