#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest -x -q tests/test_wino4_gpu.py > $O/r06_c3_tests_a.txt 2>&1; tail -3 $O/r06_c3_tests_a.txt
for sc in static dynamic static dynamic; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --tile-schedule $sc > $O/r06_c3_bench_$sc.json 2> /dev/null
  python - <<P
import json
d=json.loads([l for l in open("$O/r06_c3_bench_$sc.json") if l.startswith("{")][-1])
k=d["kernels"]; r=d["roofline"]
print("$sc", round(d["ms_per_step"],1), "wino4", round(k["conv3x3_wino4"]["ms_per_step"],2), "frac", round(r["frac"],4), "traffic GB", round((r["traffic"] or 0)/1e9,2), "wgrad4", round(k["conv3x3_wino4_wgrad"]["ms_per_step"],2), "gemm", round(k["gemm_f32"]["ms_per_step"],2))
P
done
timeout 300 python tools/exp/wino4_bench.py --only4 --sched --n 48 --iters 5 > $O/r06_w4_static_vs_dynamic_n48.txt 2>&1; grep conv $O/r06_w4_static_vs_dynamic_n48.txt; timeout 300 python tools/exp/wino4_bench.py --only4 --sched --n 16 --iters 10 > $O/r06_w4_static_vs_dynamic_n16.txt 2>&1; grep conv $O/r06_w4_static_vs_dynamic_n16.txt
for v in wg2 wg4; do timeout 400 python tools/exp/contention.py --schedule dynamic --hold 0,8,32 --lib tools/exp/_bin/libptmi355_w4w_$v.so > $O/r06_contention_$v.txt 2>&1; grep "^dynamic" $O/r06_contention_$v.txt | head -12; done
