#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest -x -q tests/test_wino4_gpu.py > $O/r06_c6_tests_a.txt 2>&1; tail -3 $O/r06_c6_tests_a.txt
timeout 900 python -m pytest -x -q -s tests/test_model_gpu.py -k "run_step_long" > $O/r06_c6_long.txt 2>&1; tail -6 $O/r06_c6_long.txt | cut -c1-2500
timeout 300 python tools/exp/wino4_bench.py --only4 --sched --n 48 --iters 5 > $O/r06_w4_static_vs_dynamic_n48.txt 2>&1; grep conv $O/r06_w4_static_vs_dynamic_n48.txt
timeout 900 python tools/exp/contention.py --hold 0,8,16,32 --schedule static,dynamic --waves 1,4 > $O/r06_contention.txt 2>&1; grep "hold" $O/r06_contention.txt
