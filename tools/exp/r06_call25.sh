#!/bin/bash
# final headline set of round 6 (kernels of the commit "first chunk with C = 0"): bench + kernel stats + idle gaps + PMC passes,
# per-layer comparison, stamps, eliminations, traffic, fuzz; then the whole GPU suite and smoke()
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=r06f; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_b16.json 2> $O/${TAG}_bench_b16.err; tail -c 200 $O/${TAG}_bench_b16.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o ${TAG} --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --pmc-traffic off > /dev/null 2> $O/${TAG}_prof.err
cp $O/${TAG}_prof/${TAG}_kernel_stats.csv $O/${TAG}_bench_b16_kernel_stats.csv
python $R/tools/exp/gaps.py $O/${TAG}_prof/${TAG}_kernel_trace.csv > $O/${TAG}_bench_b16_gpu_idle_gaps.txt 2>&1
cd $R
timeout 1200 python tools/pmc_collect.py ${TAG}_bench_b16 2>&1 | tail -2
timeout 600 python tools/exp/wino4_bench.py --only4 --p --n 16 --iters 10 > $O/${TAG}_wino4p_vs_wino4_per_layer_n16.txt 2>&1
timeout 600 python tools/exp/wino4_bench.py --only4 --p --n 48 --iters 5 > $O/${TAG}_wino4p_vs_wino4_per_layer_n48.txt 2>&1
timeout 300 python tools/exp/wino4_bench.py --only4 --p --stamps --lib tools/exp/_bin/libptmi355_wino4p_stamp.so --n 16 --layers conv1_2,conv2_2,conv3_2,conv4_2 --iters 3 --reps 1 > $O/${TAG}_wino4p_tile_stamps.txt 2>&1
for v in base noxf nolds nodma noho mfonly; do
  [ -f tools/exp/_bin/libptmi355_wino4p_$v.so ] && { echo "== $v"; timeout 200 python tools/exp/wino4_bench.py --only4 --p --lib tools/exp/_bin/libptmi355_wino4p_$v.so --layers conv3_2,conv1_2 --n 16 --iters 10 2>&1 | grep conv; }
done > $O/${TAG}_wino4p_elimination.txt 2>&1
for L in conv1_2 conv3_2 conv4_2; do timeout 300 bash tools/exp/wino4_traffic.sh "" $L 48 wino4p; done > $O/${TAG}_wino4p_traffic_per_layer_n48.txt 2>&1
timeout 400 python tools/exp/wino4_fuzz.py --seconds 240 > $O/${TAG}_wino4_fuzz.txt 2>&1; tail -2 $O/${TAG}_wino4_fuzz.txt
(time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15) > $O/${TAG}_gputests_full.txt 2>&1; tail -25 $O/${TAG}_gputests_full.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1; tail -2 $O/${TAG}_smoke.txt | cut -c1-200
