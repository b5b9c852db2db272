"""Build libptmi355.so (the HIP/CDNA4 kernels behind the train step) in-tree with hipcc for gfx950.

    python -m probabilisticteacher_amd.build_ext [--force]

No torch involvement: the library links only against libamdhip64 and exposes the plain C ABI declared
in include/ptmi355.h.  hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libptmi355.so")
SOURCES = ["abi.cpp", "conv.hip", "wino.hip", "wino4.hip", "wino4p.hip", "wino4w.hip", "p8.hip", "p8gemm.hip", "gemm.hip", "misc.hip", "boxes.hip", "nms.hip", "sort.hip", "roi_align.hip",
           "losses.hip", "augment.hip"]
# -ffp-contract=off: index-producing kernels (IoU, NMS, matcher) must evaluate fp32 expressions exactly as the
# CPU reference does.  -munsafe-fp-atomics: hardware fp32 atomic add for the ROIAlign backward scatter.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Wno-unused-result"]
# per-file extra flags.  wino4.hip: its transforms are scalar FMA sequences the SLP vectoriser would turn into v_pk_fma_f32 +
# register shuffles (slower next to MFMAs)
FILE_FLAGS = {"wino4.hip": ["-fno-slp-vectorize"], "wino4p.hip": ["-fno-slp-vectorize"], "wino4w.hip": ["-fno-slp-vectorize"]}


def _deps_mtime():
    paths = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    paths.append(os.path.join(HERE, "..", "include", "ptmi355.h"))
    return max(os.path.getmtime(p) for p in paths)


def _compile(src):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    hdr_m = max(os.path.getmtime(os.path.join(CSRC, "common.h")),
                os.path.getmtime(os.path.join(HERE, "..", "include", "ptmi355.h")))
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), hdr_m):
        return obj
    cmd = ["hipcc"] + FLAGS + FILE_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", path, "-o", obj]
    subprocess.check_call(cmd)
    return obj


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _deps_mtime():
        return LIB
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(_compile, SOURCES))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
