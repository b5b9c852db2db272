#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest -x -q tests/test_wino4_gpu.py -k "wino4p" > $O/r06_c9_tests_p.txt 2>&1; tail -3 $O/r06_c9_tests_p.txt | cut -c1-220
timeout 300 python tools/exp/wino4_bench.py --only4 --p --n 48 --iters 5 > $O/r06_w4p_vs_w4_n48.txt 2>&1; grep conv $O/r06_w4p_vs_w4_n48.txt
timeout 300 python tools/exp/wino4_bench.py --only4 --p --n 16 --iters 10 > $O/r06_w4p_vs_w4_n16.txt 2>&1; grep conv $O/r06_w4p_vs_w4_n16.txt
timeout 300 python tools/exp/wino4_bench.py --only4 --p --stamps --lib tools/exp/_bin/libptmi355_wino4p_stamp.so --n 16 --layers conv1_2,conv3_2 --iters 3 --reps 1 2>&1 | grep -v amdgpu
