#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest -x -q tests/test_wino4_gpu.py -k "wino4p" > $O/r06_c7_tests_p.txt 2>&1; tail -25 $O/r06_c7_tests_p.txt | cut -c1-220
timeout 300 python tools/exp/wino4_bench.py --only4 --p --n 16 --iters 10 > $O/r06_w4p_vs_w4_n16.txt 2>&1; grep conv $O/r06_w4p_vs_w4_n16.txt
timeout 300 python tools/exp/wino4_bench.py --only4 --p --n 48 --iters 5 > $O/r06_w4p_vs_w4_n48.txt 2>&1; grep conv $O/r06_w4p_vs_w4_n48.txt
timeout 900 python -m pytest -x -q -s tests/test_model_gpu.py -k "run_step_long" > $O/r06_c7_long.txt 2>&1; tail -6 $O/r06_c7_long.txt | cut -c1-2500
