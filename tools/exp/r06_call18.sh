#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_p8_gpu.py tests/test_wino_gpu.py tests/test_wino4_gpu.py -m gpu -q -s --durations=10 -k "loss_weight or test_ops_gpu or test_p8_gpu or test_wino_gpu or test_wino4_gpu") > $O/r06_gputests_rest.txt 2>&1; tail -30 $O/r06_gputests_rest.txt | cut -c1-220
timeout 900 python tools/exp/contention.py --amp --hold 0,8,32 --schedule static --waves 1,4 --p8-waves 1,16 > $O/r06_contention_amp.txt 2>&1; tail -20 $O/r06_contention_amp.txt | cut -c1-200
