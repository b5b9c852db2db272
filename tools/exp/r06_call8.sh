#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
echo "== wino4 (round 5 structure)" > $O/r06_w4p_stamps.txt
timeout 300 python tools/exp/wino4_bench.py --only4 --stamps --lib tools/exp/_bin/libptmi355_w4_stamp.so --n 16 --layers conv1_2,conv2_2,conv3_2,conv4_2 --iters 3 --reps 1 >> $O/r06_w4p_stamps.txt 2>&1
echo "== wino4p (round 6: positions split)" >> $O/r06_w4p_stamps.txt
timeout 300 python tools/exp/wino4_bench.py --only4 --p --stamps --lib tools/exp/_bin/libptmi355_wino4p_stamp.so --n 16 --layers conv1_2,conv2_2,conv3_2,conv4_2 --iters 3 --reps 1 >> $O/r06_w4p_stamps.txt 2>&1
grep -v amdgpu $O/r06_w4p_stamps.txt
timeout 900 python -m pytest -x -q -s tests/test_model_gpu.py -k "run_step_long" > $O/r06_c8_long.txt 2>&1; tail -4 $O/r06_c8_long.txt | cut -c1-2500
