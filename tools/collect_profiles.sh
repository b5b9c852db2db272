#!/bin/bash
# Round profile set, on the GPU box:  bash tools/collect_profiles.sh r03   -> gpurun_out/<tag>_* (copy what is judged into profiles/)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_b16.json 2> gpurun_out/${TAG}_bench_b16.err
tail -c 300 gpurun_out/${TAG}_bench_b16.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o ${TAG} --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --pmc-traffic off > $R/gpurun_out/${TAG}_bench_b16_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.err
cp $R/gpurun_out/${TAG}_prof/${TAG}_kernel_stats.csv $R/gpurun_out/${TAG}_bench_b16_kernel_stats.csv
cd $R
timeout 1200 python tools/pmc_collect.py ${TAG}_bench_b16 2>&1 | tail -2
timeout 600 python tools/kbench.py --n 16 --iters 3 > gpurun_out/${TAG}_kbench_per_layer_n16.txt 2>&1
timeout 600 python tools/kbench.py --n 48 --which conv,wgrad --iters 2 > gpurun_out/${TAG}_kbench_per_layer_n48.txt 2>&1
timeout 900 python bench.py --amp > gpurun_out/${TAG}_bench_b16_amp.json 2> gpurun_out/${TAG}_bench_b16_amp.err
ls gpurun_out | head -40
