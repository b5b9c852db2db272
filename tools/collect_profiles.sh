#!/bin/bash
# Round profile set, on the GPU box:  bash tools/collect_profiles.sh r06   -> gpurun_out/<tag>_* (copy what is judged into profiles/)
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
# the headline line (fp32, 16 + 16), with traffic measured in the run and the CPU baseline
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_b16.json 2> gpurun_out/${TAG}_bench_b16.err
tail -c 300 gpurun_out/${TAG}_bench_b16.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o ${TAG} --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --pmc-traffic off > $R/gpurun_out/${TAG}_bench_b16_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.err
cp $R/gpurun_out/${TAG}_prof/${TAG}_kernel_stats.csv $R/gpurun_out/${TAG}_bench_b16_kernel_stats.csv
python $R/tools/exp/gaps.py $R/gpurun_out/${TAG}_prof/${TAG}_kernel_trace.csv > $R/gpurun_out/${TAG}_bench_b16_gpu_idle_gaps.txt 2>&1
cd $R
timeout 1200 python tools/pmc_collect.py ${TAG}_bench_b16 2>&1 | tail -2
timeout 600 python tools/kbench.py --n 16 --iters 3 > gpurun_out/${TAG}_kbench_per_layer_n16.txt 2>&1
timeout 600 python tools/kbench.py --n 48 --which conv,wgrad --iters 2 > gpurun_out/${TAG}_kbench_per_layer_n48.txt 2>&1
# SOLVER.AMP.ENABLED (bf16 storage kernels): bench line (traffic measured in the run), kernel stats, per-layer kernels, PMC of the two MFMA kernels
timeout 900 python bench.py --amp --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_b16_amp.json 2> gpurun_out/${TAG}_bench_b16_amp.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_amp -o ${TAG}amp --output-format csv -- python $R/bench.py --amp --steps 2 --warmup 1 --no-cpu-baseline --pmc-traffic off > /dev/null 2> $R/gpurun_out/${TAG}_rocprof_amp.err
cp $R/gpurun_out/${TAG}_prof_amp/${TAG}amp_kernel_stats.csv $R/gpurun_out/${TAG}_bench_b16_amp_kernel_stats.csv
python $R/tools/exp/gaps.py $R/gpurun_out/${TAG}_prof_amp/${TAG}amp_kernel_trace.csv > $R/gpurun_out/${TAG}_bench_b16_amp_gpu_idle_gaps.txt 2>&1
cd $R
timeout 600 python tools/kbench_p8.py --n 16 --which conv,dgrad,wgrad,pool --iters 5 > gpurun_out/${TAG}_kbench_p8_per_layer_n16_bf16.txt 2>&1
timeout 600 python tools/kbench_p8.py --n 48 --which conv,dgrad,wgrad --iters 3 > gpurun_out/${TAG}_kbench_p8_per_layer_n48_bf16.txt 2>&1
timeout 600 python tools/kbench_p8.py --n 16 --which gemm --iters 3 > gpurun_out/${TAG}_kbench_p8_fc1_gemm_bf16.txt 2>&1
timeout 600 bash tools/exp/p8_pmc.sh conv3_2 conv 48 > gpurun_out/${TAG}_p8_conv3_2_fwd_n48_pmc.txt 2>&1
timeout 600 bash tools/exp/p8_pmc.sh conv3_2 wgrad 48 > gpurun_out/${TAG}_p8_conv3_2_wgrad_n48_pmc.txt 2>&1
# round 5: the F(4x4,3x3) kernels alone -- per-layer times against the F(2x2,3x3) kernels, per-tile phase stamps + shader clock,
# elimination variants (tools/exp/make_wino4_variant.py must have been run in the dev container: tools/exp/_bin travels), HBM-side traffic
timeout 600 python tools/exp/wino4_bench.py --n 16 --iters 10 > gpurun_out/${TAG}_wino4_vs_wino_per_layer_n16.txt 2>&1
timeout 600 python tools/exp/wino4_bench.py --n 48 --iters 5 > gpurun_out/${TAG}_wino4_vs_wino_per_layer_n48.txt 2>&1
timeout 600 python tools/exp/wino4w_bench.py --n 48 > gpurun_out/${TAG}_wino4_wgrad_vs_wino_wgrad_n48.txt 2>&1
if [ -f tools/exp/_bin/libptmi355_w4_stamp.so ]; then
  timeout 300 python tools/exp/wino4_bench.py --only4 --stamps --lib tools/exp/_bin/libptmi355_w4_stamp.so --n 16 --layers conv1_2,conv2_2,conv3_2,conv4_2 --iters 3 --reps 1 > gpurun_out/${TAG}_wino4_tile_stamps.txt 2>&1
  for v in base noxf nolds nodma noho mfonly; do
    [ -f tools/exp/_bin/libptmi355_w4_$v.so ] && { echo "== $v"; timeout 200 python tools/exp/wino4_bench.py --only4 --lib tools/exp/_bin/libptmi355_w4_$v.so --layers conv3_2,conv1_2 --n 16 --iters 10 2>&1 | grep conv; }
  done > gpurun_out/${TAG}_wino4_elimination.txt 2>&1
fi
for v in base noxf nord nodma noho mfonly; do
  [ -f tools/exp/_bin/libptmi355_w4w_$v.so ] && { echo "== $v"; timeout 200 python tools/exp/wino4w_bench.py --lib tools/exp/_bin/libptmi355_w4w_$v.so --layers conv3_2,conv4_2 --n 16 2>&1 | grep conv; }
done > gpurun_out/${TAG}_wino4_wgrad_elimination.txt 2>&1
for L in conv1_2 conv3_2 conv4_2; do timeout 300 bash tools/exp/wino4_traffic.sh "" $L 48; done > gpurun_out/${TAG}_wino4_traffic_per_layer_n48.txt 2>&1
# round 6: the position-split kernel (csrc/wino4p.hip, the default) against round 5's (csrc/wino4.hip), per layer, same process; its
# tile stamps; the static walk against the dynamic schedule per layer; the CU-contention experiment; which XCD a workgroup runs on
timeout 600 python tools/exp/wino4_bench.py --only4 --p --n 16 --iters 10 > gpurun_out/${TAG}_wino4p_vs_wino4_per_layer_n16.txt 2>&1
timeout 600 python tools/exp/wino4_bench.py --only4 --p --n 48 --iters 5 > gpurun_out/${TAG}_wino4p_vs_wino4_per_layer_n48.txt 2>&1
if [ -f tools/exp/_bin/libptmi355_wino4p_stamp.so ]; then
  timeout 300 python tools/exp/wino4_bench.py --only4 --p --stamps --lib tools/exp/_bin/libptmi355_wino4p_stamp.so --n 16 --layers conv1_2,conv2_2,conv3_2,conv4_2 --iters 3 --reps 1 > gpurun_out/${TAG}_wino4p_tile_stamps.txt 2>&1
fi
for v in base noxf nolds nodma noho mfonly; do
  [ -f tools/exp/_bin/libptmi355_wino4p_$v.so ] && { echo "== $v"; timeout 200 python tools/exp/wino4_bench.py --only4 --p --lib tools/exp/_bin/libptmi355_wino4p_$v.so --layers conv3_2,conv1_2 --n 16 --iters 10 2>&1 | grep conv; }
done > gpurun_out/${TAG}_wino4p_elimination.txt 2>&1
timeout 600 python tools/exp/wino4_bench.py --only4 --sched --n 48 --iters 5 > gpurun_out/${TAG}_wino4_static_vs_dynamic_n48.txt 2>&1
[ -f tools/exp/_bin/libptmi355_w4_r05.so ] && { for rep in 1 2; do for L in "" "--lib tools/exp/_bin/libptmi355_w4_r05.so"; do echo "== ${L:-product (round 6, static instantiation)} rep $rep"; timeout 300 python tools/exp/wino4_bench.py --only4 --n 48 --iters 5 --layers conv1_2,conv2_2,conv3_2,conv4_2,conv5_1 $L 2>&1 | grep conv; done; done; } > gpurun_out/${TAG}_wino4_r05_vs_r06_n48.txt 2>&1
timeout 900 python tools/exp/contention.py --hold 0,8,16,32 --schedule static,dynamic --waves 1,4 > gpurun_out/${TAG}_contention.txt 2>&1
[ -x tools/exp/_bin/xcc_probe ] && tools/exp/_bin/xcc_probe > gpurun_out/${TAG}_xcc_probe.txt 2>&1
for L in conv1_2 conv3_2 conv4_2; do timeout 300 bash tools/exp/wino4_traffic.sh "" $L 48 wino4p; done > gpurun_out/${TAG}_wino4p_traffic_per_layer_n48.txt 2>&1
# ROIAlign at the step's launch shapes / ROI extents, host-boundness of the step
timeout 300 python tools/exp/roi_bench.py > gpurun_out/${TAG}_roi_align_step_shapes.txt 2>&1
timeout 600 python tools/exp/host_bound.py > gpurun_out/${TAG}_host_bound_fp32.txt 2>&1
timeout 600 python tools/exp/host_bound.py --amp > gpurun_out/${TAG}_host_bound_amp.txt 2>&1
# BASELINE configs[1] at its own batch (8 images = 4 labelled x 2 views)
timeout 600 python bench.py --student-only --per-gpu-batch 4 --steps 20 --warmup 5 --pmc-traffic off > gpurun_out/${TAG}_bench_config1_student_only_b8.json 2> /dev/null
# BASELINE configs[3] per-GPU share (8 + 8) on one GPU: the N = 1 point of its weak-scaling curve
timeout 600 python bench.py --config3 --steps 20 --warmup 5 --pmc-traffic off --no-cpu-baseline > gpurun_out/${TAG}_bench_config3_per_gpu_share_n1.json 2> /dev/null
ls gpurun_out | head -60
