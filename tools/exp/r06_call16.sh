#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest -q tests/test_ops_gpu.py -k "efl or soft" > $O/r06_c16_efl.txt 2>&1; tail -5 $O/r06_c16_efl.txt | cut -c1-300
(time timeout 3000 python -m pytest -q -s tests/test_config4_gpu.py -k "loss_curves") > $O/r06_loss_curves_v5.txt 2>&1; grep -c "ended by" $O/r06_loss_curves_v5.txt; tail -6 $O/r06_loss_curves_v5.txt | cut -c1-200
