set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 2> gpurun_out/bench_b16.err | tail -1 > gpurun_out/r02_bench_b16.json
python bench.py --student-only --per-gpu-batch 4 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r02_bench_config1_student_only_b8.json
python bench.py --amp --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r02_bench_b16_amp.json
python tools/kbench.py --n 16 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_kbench_per_layer_n16.txt
python tools/kbench.py --n 48 --which conv,wgrad 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_kbench_per_layer_n48.txt
python tools/kbench.py --n 16 --bf16 --which conv,wgrad,gemm 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_kbench_per_layer_n16_bf16.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_b16 -o b16 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_b16.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_amp -o amp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --amp --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_amp.log 2>&1)
find gpurun_out/prof_b16 gpurun_out/prof_amp -name "*kernel_stats.csv" | head
find gpurun_out/prof_b16 gpurun_out/prof_amp -name "*kernel_trace.csv" -delete
find gpurun_out/prof_b16 gpurun_out/prof_amp -name "*.db" -delete
cut -c1-400 gpurun_out/r02_bench_b16.json
tail -1 gpurun_out/bench_b16.err
