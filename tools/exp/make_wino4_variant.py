#!/usr/bin/env python
"""Timing variants of the F(4x4,3x3) kernel (results WRONG by design): copies of csrc/wino4.hip with the lines tagged
`// [x4:<part>]` removed, built into tools/exp/_bin/libptmi355_w4_<name>.so (the other objects are the product's).

    python tools/exp/make_wino4_variant.py base none   noxf xf   nowr wr   noar ar   nodma dma   noho ho   nomf mf   mfonly xf,wr,ar,dma,ho
    (on the GPU box)  python tools/exp/wino4_bench.py --only4 --lib tools/exp/_bin/libptmi355_w4_noxf.so --layers conv3_2

parts: mf = the MFMAs, xf = the input transforms' FMAs, wr = window reads, ar = A-operand reads, dma = the LDS-DMA instructions,
ho = the hand-over (counted vmcnt, fix-ups, barrier).  W4FILE=wino4w: the weight-gradient kernel (parts mf, xf, rd = raw operand
reads, dma, ho) -> libptmi355_w4w_<name>.so.  The `else` of a removed `if constexpr` pair goes with it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "probabilisticteacher_amd", "csrc", "wino4.hip")
BIN = os.path.join(ROOT, "tools", "exp", "_bin")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-Wno-unused-result",
         "-Wno-uninitialized", "-Wno-sometimes-uninitialized"]


def main():
    os.makedirs(BIN, exist_ok=True)
    subprocess.check_call([sys.executable, "-m", "probabilisticteacher_amd.build_ext"], cwd=ROOT, stdout=subprocess.DEVNULL)
    objs = [os.path.join(ROOT, "probabilisticteacher_amd", "_build", f)
            for f in os.listdir(os.path.join(ROOT, "probabilisticteacher_amd", "_build")) if f.endswith(".o") and f != "wino4.o"]
    args = sys.argv[1:]
    extra = []
    while args and args[0].startswith("-"):
        extra.append(args.pop(0))
    which = os.environ.get("W4FILE", "wino4")       # wino4 (forward / dgrad) or wino4w (weight gradient)
    objs = [o for o in objs if os.path.basename(o) != which + ".o"] + ([os.path.join(ROOT, "probabilisticteacher_amd", "_build", "wino4.o")] if which != "wino4" else [])
    src = open(os.environ.get("W4SRC", os.path.join(ROOT, "probabilisticteacher_amd", "csrc", which + ".hip"))).read().splitlines()
    for name, parts in zip(args[0::2], args[1::2]):
        drop = set() if parts == "none" else set(parts.split(","))
        out = [ln for ln in src if not any(f"[x4:{p}]" in ln for p in drop)]
        if "stamp" in drop:
            # `// [x4@tK]` markers become s_memtime stamps of wave 0 of workgroup 0: slot 8 * tile + K of the int64 buffer passed as
            # mask_ref (run with an epilogue other than 3; tools/exp/wino4_bench.py --stamps)
            def stamp(ln):
                if "[x4@t" not in ln:
                    return ln
                k = int(ln.split("[x4@t")[1].split("]")[0])
                return (f"        if (blockIdx.x == 0 && tid == 0 && x4_tile < 64) {{ unsigned long long t_; asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t_)); "
                        f"((unsigned long long*)mref)[8 * x4_tile + {k}] = t_; " +
                        ("asm volatile(\"s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t_)); ((unsigned long long*)mref)[8 * x4_tile + 5] = t_; " if k == 0 else "") +
                        "}" + ("  ++x4_tile;" if k == 4 else ""))
            out = [stamp(ln) for ln in out]
            out = [ln.replace("    int vid = blockIdx.x;", "    int vid = blockIdx.x; int x4_tile = 0;") for ln in out]
        tmp = os.path.join(ROOT, "probabilisticteacher_amd", "csrc", f"_w4_{name}.hip")
        open(tmp, "w").write("\n".join(out) + "\n")
        try:
            obj = os.path.join(BIN, f"w4_{name}.o")
            subprocess.check_call(["hipcc"] + FLAGS + extra + ["-c", tmp, "-o", obj])
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                                   os.path.join(BIN, f"libptmi355_{ {'wino4': 'w4', 'wino4w': 'w4w'}.get(which, which) }_{name}.so")] + objs + [obj])
        finally:
            os.remove(tmp)
        print("built", name, "without", sorted(drop))


if __name__ == "__main__":
    main()
