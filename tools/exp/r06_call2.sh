#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
tools/exp/_bin/xcc_probe > $O/r06_xcc_probe.txt 2>&1; cat $O/r06_xcc_probe.txt
timeout 300 python tools/exp/w4w_cin32_debug.py > $O/r06_w4w_cin32_debug.txt 2>&1; tail -14 $O/r06_w4w_cin32_debug.txt
timeout 600 python -m pytest -x -q tests/test_wino_gpu.py tests/test_wino4_gpu.py -k "routing or contention" > $O/r06_c2_tests_a.txt 2>&1; tail -3 $O/r06_c2_tests_a.txt
timeout 600 python tools/exp/contention.py > $O/r06_contention_base.txt 2>&1; grep "hold\|^static\|^dynamic" $O/r06_contention_base.txt | head -30
for v in wg2 wg4; do timeout 400 python tools/exp/contention.py --schedule dynamic --lib tools/exp/_bin/libptmi355_w4w_$v.so > $O/r06_contention_$v.txt 2>&1; grep "hold\|^dynamic" $O/r06_contention_$v.txt | head -12; done
timeout 1200 python -m pytest -x -q -s tests/test_config4_gpu.py -k "trained_weights" > $O/r06_c2_trained.txt 2>&1; tail -40 $O/r06_c2_trained.txt
