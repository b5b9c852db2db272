#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest -q tests/test_ddp_gpu.py tests/test_model_gpu.py tests/test_p8_gpu.py -k "bitwise_identity or loss_weight_quirk or oversize_fallback or run_step_long" > $O/r06_c12_tests.txt 2>&1; tail -4 $O/r06_c12_tests.txt | cut -c1-300
bash tools/collect_profiles.sh r06 > $O/r06_collect.log 2>&1; tail -5 $O/r06_collect.log
