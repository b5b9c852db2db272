#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06b_bench_b16.json 2> $O/r06b_bench_b16.err; tail -c 400 $O/r06b_bench_b16.json; tail -2 $O/r06b_bench_b16.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06b_prof -o r06b --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --pmc-traffic off > $R/gpurun_out/r06b_bench_b16_under_rocprof.json 2> $R/gpurun_out/r06b_rocprof.err
cp $R/gpurun_out/r06b_prof/r06b_kernel_stats.csv $R/gpurun_out/r06b_bench_b16_kernel_stats.csv
python $R/tools/exp/gaps.py $R/gpurun_out/r06b_prof/r06b_kernel_trace.csv > $R/gpurun_out/r06b_bench_b16_gpu_idle_gaps.txt 2>&1
cd $R
timeout 1200 python tools/pmc_collect.py r06b_bench_b16 2>&1 | tail -2
(time timeout 3000 python -m pytest -q -s tests/test_config4_gpu.py -k "loss_curves") > $O/r06_loss_curves_v5.txt 2>&1; grep -v "^\s*$" $O/r06_loss_curves_v5.txt | tail -40 | cut -c1-420
