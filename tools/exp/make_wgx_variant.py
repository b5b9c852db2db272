#!/usr/bin/env python
"""Write an INSTRUMENTED copy of csrc/wino.hip with the weight-gradient kernel's elimination switches (-DWGX=<mask>:
1 no DMA, 2 no transforms, 4 no LDS reads, 8 no hand-over, 16 no MFMA -- results WRONG by design when non-zero).

    python tools/exp/make_wgx_variant.py tools/exp/_snap/wino_wgx.hip
    WINO_FLAGS=-DWGX=2 tools/exp/wino_variants.sh notransform=tools/exp/_snap/wino_wgx.hip

The product source carries none of these switches (round 4: they were moved here); this script re-inserts them by exact
string replacement and fails if the product code has drifted from the anchors."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "probabilisticteacher_amd", "csrc", "wino.hip")

EDITS = [
    ("constexpr int GWC = 64;", "#ifndef WGX\n#define WGX 0\n#endif\nconstexpr int GWC = 64;"),
    ("                acc[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa(M, P), opb(M, P), acc[P], 0, 0, 0);",
     "                if (!(WGX & 16)) acc[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa(M, P), opb(M, P), acc[P], 0, 0, 0);"),
    ("                    wino_vmwait0();\n                    fixup(cur ^ 1, fix_h);\n"
     "                    asm volatile(\"s_waitcnt lgkmcnt(0)\\n\\ts_barrier\" ::: \"memory\");\n",
     "                    if (!(WGX & 8)) {\n                    wino_vmwait0();\n                    fixup(cur ^ 1, fix_h);\n"
     "                    asm volatile(\"s_waitcnt lgkmcnt(0)\\n\\ts_barrier\" ::: \"memory\");\n                    }\n"),
    ("if constexpr (P < 10) raw_read(src, wb, KR, RA[M], RB[M], P);",
     "if constexpr (P < 10) { if (!(WGX & 4)) raw_read(src, wb, KR, RA[M], RB[M], P); }"),
    ("                        fetch_piece(part * 4 + sub, late ? cur : cur ^ 1);",
     "                        if (!(WGX & 1)) fetch_piece(part * 4 + sub, late ? cur : cur ^ 1);"),
    ("                if constexpr (P == 2) xform_a_rows(RA[O], WR[O]);", "                if (!(WGX & 2)) {\n                if constexpr (P == 2) xform_a_rows(RA[O], WR[O]);"),
    ("                if constexpr (P >= 10 && P <= 13) xform_b_col(VX[O], VY[O], P - 10);\n",
     "                if constexpr (P >= 10 && P <= 13) xform_b_col(VX[O], VY[O], P - 10);\n                }\n"),
]


def main():
    out = sys.argv[1]
    s = open(SRC).read()
    for old, new in EDITS:
        assert s.count(old) == 1, f"anchor not unique / missing in csrc/wino.hip: {old[:60]!r}"
        s = s.replace(old, new)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    open(out, "w").write(s)
    print(out)


if __name__ == "__main__":
    main()
