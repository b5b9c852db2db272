#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
(time timeout 3000 python -m pytest tests -m gpu -q -k "not loss_curves" --deselect tests/test_augment_gpu.py --deselect tests/test_baseline_size_gpu.py) > $O/r06_gputests_c11.txt 2>&1; tail -25 $O/r06_gputests_c11.txt | cut -c1-300
