#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06_bench_c10.json 2> $O/r06_bench_c10.err; tail -c 600 $O/r06_bench_c10.json; tail -2 $O/r06_bench_c10.err
python - <<P
import json
d=json.loads([l for l in open("$O/r06_bench_c10.json") if l.startswith("{")][-1])
k=d["kernels"]; r=d["roofline"]
print("bench", round(d["value"],2), round(d["ms_per_step"],1), "wino4", round(k["conv3x3_wino4"]["ms_per_step"],2), "frac", round(r["frac"],4), "traffic GB", round((r["traffic"] or 0)/1e9,2), "wgrad4", round(k["conv3x3_wino4_wgrad"]["ms_per_step"],2), "gemm", round(k["gemm_f32"]["ms_per_step"],2))
P
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/r06_gputests_c10.txt 2>&1; tail -15 $O/r06_gputests_c10.txt | cut -c1-300
