#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_ddp_gpu.py tests/test_wino4_gpu.py tests/test_p8_gpu.py -m gpu -q -x -k "ddp or waves") > $O/r06_policy_tests.txt 2>&1; tail -6 $O/r06_policy_tests.txt | cut -c1-200
