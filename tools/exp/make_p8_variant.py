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
    ("                const ptmi_bf16x8 A0 = rd(base + a_base + s * 256), A1 = rd(base + a_base + 4 * G_DYP + s * 256);\n"
     "                ptmi_bf16x8 B[5];\n#pragma unroll\n"
     "                for (int t = 0; t < 5; ++t) B[t] = rd(base + b_base[t] + ((s >> 1) * 34 + 16 * (s & 1)) * 16);",
     "                ptmi_bf16x8 A0 = ones, A1 = ones;\n                ptmi_bf16x8 B[5];\n"
     "                if (!(P8X & 32)) { A0 = rd(base + a_base + s * 256); A1 = rd(base + a_base + 4 * G_DYP + s * 256); }\n#pragma unroll\n"
     "                for (int t = 0; t < 5; ++t) B[t] = (P8X & 32) ? ones : rd(base + b_base[t] + ((s >> 1) * 34 + 16 * (s & 1)) * 16);"),
    ("        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        asm volatile(\"s_waitcnt lgkmcnt(0)\\n\\ts_barrier\" ::: \"memory\");\n    }\n\n    // ---- partials",
     "        if (!(P8X & 64)) {\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        asm volatile(\"s_waitcnt lgkmcnt(0)\\n\\ts_barrier\" ::: \"memory\");\n        }\n    }\n\n    // ---- partials"),
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
