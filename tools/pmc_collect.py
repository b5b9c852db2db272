#!/usr/bin/env python
"""Collect rocprofv3 PMC counters of the bench command, in separate passes (one counter group per run, --kernel-trace
only, as gpurun requires), and aggregate them per kernel into profiles/<tag>_pmc_by_kernel.json.

    python tools/pmc_collect.py r02_bench_b16 [bench flags ...]      (on the GPU box; writes under gpurun_out/ and profiles/)

FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3 derived counters); FETCH_SIZE under-reports wide coalesced reads by 2x on
gfx950 (MI355X_MICROARCH.md, HBM section) -- consumers double it."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ["FETCH_SIZE", "GRBM_GUI_ACTIVE"],
    ["WRITE_SIZE"],
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
    ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_MOPS_F32"],
]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    s = m.group(1) if m else name
    return s[:90]


def main():
    tag = sys.argv[1]
    bench_flags = (sys.argv[2:] or ["--steps", "2", "--warmup", "1", "--no-cpu-baseline"]) + ["--pmc-traffic", "off"]
    os.environ["TMPDIR"] = "/tmp"
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(int)
    for i, counters in enumerate(PASSES):
        d = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc{i}")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "-d", d, "-o", f"p{i}", "--output-format", "csv", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), *bench_flags]
        print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print(r.stdout[-1500:], flush=True)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        seen = set()
        for f in files:
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                if i == 0 and (row["Dispatch_Id"], f) not in seen:
                    seen.add((row["Dispatch_Id"], f))
                    disp[k] += 1
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, stdout=subprocess.PIPE, text=True).stdout.strip()
    ours = {k: dict(v, dispatches=disp[k]) for k, v in agg.items() if not k.startswith("at::") and "rocprim" not in k and disp[k]}
    other = {"dispatches": sum(disp[k] for k in agg if k not in ours),
             "FETCH_SIZE": sum(v.get("FETCH_SIZE", 0.0) for k, v in agg.items() if k not in ours),
             "WRITE_SIZE": sum(v.get("WRITE_SIZE", 0.0) for k, v in agg.items() if k not in ours)}
    out = {"git_head": head or os.environ.get("GRAFT_HEAD", "working tree"), "command": "bench.py " + " ".join(bench_flags),
           "passes": PASSES, "units": {"FETCH_SIZE": "KB (x2 on gfx950 for wide reads)", "WRITE_SIZE": "KB"},
           "kernels": ours, "aten_and_rocprim_kernels": other}
    dst = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_by_kernel.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst)


if __name__ == "__main__":
    main()
