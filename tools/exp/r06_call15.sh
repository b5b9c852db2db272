#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
(time timeout 3000 python -m pytest -q -s tests/test_config4_gpu.py -k "loss_curves_vs_committed") > $O/r06_loss_curves_v5_fp32.txt 2>&1; grep -v "^\s*$" $O/r06_loss_curves_v5_fp32.txt | tail -40 | cut -c1-500
