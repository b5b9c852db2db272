#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
for rep in 1 2; do
for L in "" "--lib tools/exp/_bin/libptmi355_w4_r05.so"; do
  echo "== ${L:-product (round 6)} rep $rep"; timeout 300 python tools/exp/wino4_bench.py --only4 --n 48 --iters 5 --layers conv1_2,conv2_2,conv3_2,conv4_2,conv5_1 $L 2>&1 | grep conv
done; done > $O/r06_w4_r05_vs_r06_n48.txt 2>&1; cat $O/r06_w4_r05_vs_r06_n48.txt
for L in conv3_2 conv4_2 conv1_2; do timeout 300 bash tools/exp/wino4_traffic.sh "" $L 48; timeout 300 bash tools/exp/wino4_traffic.sh tools/exp/_bin/libptmi355_w4_r05.so $L 48; done > $O/r06_w4_traffic_r05_vs_r06.txt 2>&1; cat $O/r06_w4_traffic_r05_vs_r06.txt
timeout 900 python -m pytest -x -q -s tests/test_model_gpu.py -k "run_step_long" > $O/r06_c4_long.txt 2>&1; tail -12 $O/r06_c4_long.txt | cut -c1-1500
