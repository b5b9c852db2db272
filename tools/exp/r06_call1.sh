#!/bin/bash
# round 6, GPU call 1: new tests, contention experiment, bench with legs
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
(free -g; nproc) > $O/r06_box.txt 2>&1
timeout 900 python -m pytest -x -q tests/test_wino4_gpu.py tests/test_wino_gpu.py tests/test_abi.py "tests/test_ops_gpu.py::test_segsort_host_lengths_guard" tests/test_p8_gpu.py -k "sched or routing or abi or segsort or fallback or rejects or dynamic or equals_direct" > $O/r06_c1_tests_a.txt 2>&1; tail -5 $O/r06_c1_tests_a.txt
timeout 900 python -m pytest -x -q -s tests/test_baseline_size_gpu.py -k "n48" > $O/r06_c1_tests_b.txt 2>&1; tail -5 $O/r06_c1_tests_b.txt
timeout 600 python tools/exp/contention.py > $O/r06_contention_base.txt 2>&1; tail -30 $O/r06_contention_base.txt
for v in wg2 wg4; do timeout 400 python tools/exp/contention.py --schedule dynamic --lib tools/exp/_bin/libptmi355_w4w_$v.so > $O/r06_contention_$v.txt 2>&1; grep "hold" $O/r06_contention_$v.txt | head -8; done
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06_bench_c1.json 2> $O/r06_bench_c1.err; tail -c 1500 $O/r06_bench_c1.json; tail -3 $O/r06_bench_c1.err
/usr/bin/time -v timeout 900 python -m pytest -x -q -s tests/test_baseline_size_gpu.py -k "b8_plus_8" > $O/r06_c1_tests_c.txt 2>&1; grep -i "maximum resident\|passed\|failed\|Elapsed" $O/r06_c1_tests_c.txt
