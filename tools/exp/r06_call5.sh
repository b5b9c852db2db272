#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest -x -q tests/test_wino4_gpu.py > $O/r06_c5_tests_a.txt 2>&1; tail -3 $O/r06_c5_tests_a.txt
PTMI_TEST_VERBOSE=1 timeout 900 python -m pytest -x -q -s tests/test_model_gpu.py -k "run_step_long" > $O/r06_c5_long.txt 2>&1; grep -v "^\[close\]" $O/r06_c5_long.txt | tail -12 | cut -c1-1800; grep "^\[close\]" $O/r06_c5_long.txt | awk '{print $NF, $0}' | sort -g | tail -8
timeout 900 python tools/exp/contention.py --hold 0,8,16,32 --schedule static,dynamic --waves 1,4 > $O/r06_contention.txt 2>&1; grep -v "^{" $O/r06_contention.txt | tail -40
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/r06_c5_bench.json 2> /dev/null
python - <<P
import json
d=json.loads([l for l in open("$O/r06_c5_bench.json") if l.startswith("{")][-1])
k=d["kernels"]; r=d["roofline"]
print("bench", round(d["ms_per_step"],1), "wino4", round(k["conv3x3_wino4"]["ms_per_step"],2), "frac", round(r["frac"],4), "traffic GB", round((r["traffic"] or 0)/1e9,2), "wgrad4", round(k["conv3x3_wino4_wgrad"]["ms_per_step"],2), "gemm", round(k["gemm_f32"]["ms_per_step"],2))
P
timeout 600 python tools/hostprof.py 16 > $O/r06_hostprof_fp32.txt 2>&1; tail -40 $O/r06_hostprof_fp32.txt | cut -c1-200
timeout 600 python tools/synctrace.py > $O/r06_synctrace.txt 2>&1; cat $O/r06_synctrace.txt | tail -20
