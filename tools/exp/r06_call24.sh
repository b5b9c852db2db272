#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
(time timeout 900 python -m pytest tests/test_wino4_gpu.py tests/test_baseline_size_gpu.py -m gpu -q -x -k "not 16_plus_16 and not loss_curve") > $O/r06_first_chunk_tests.txt 2>&1; tail -8 $O/r06_first_chunk_tests.txt | cut -c1-200
python bench.py --steps 20 --warmup 5 > $O/r06_bench_first_chunk.json 2> $O/r06_bench_first_chunk.err; python - <<'P'
import json
d=json.loads(open("gpurun_out/r06_bench_first_chunk.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"], d["kernels"]["conv3x3_wino4"], d["legs"])
P
