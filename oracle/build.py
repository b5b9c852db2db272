"""Build the plain-C part of the CPU oracle (TEST INFRASTRUCTURE) with gcc.

    python -m oracle.build        ->  oracle/_build/libptoracle.so
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "ref_ops.c")
SRCS = [SRC, os.path.join(HERE, "csrc", "ref_aug.c")]
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libptoracle.so")


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= max(os.path.getmtime(f) for f in SRCS)):
        return OUT
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", *SRCS, "-o", OUT, "-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
