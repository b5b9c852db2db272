#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python tools/soak.py 600 > $O/r06f_soak_600.txt 2>&1; tail -6 $O/r06f_soak_600.txt | cut -c1-200
timeout 900 python tools/exp/contention.py --hold 0,8,32 --schedule static,dynamic --waves 1,4 > $O/r06f_contention.txt 2>&1; grep -v "^{" $O/r06f_contention.txt | head -16 | cut -c1-200
timeout 600 python bench.py --config3 --steps 20 --warmup 5 --pmc-traffic off --no-cpu-baseline > $O/r06f_bench_config3_per_gpu_share_n1.json 2> /dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06f_bench_config3_per_gpu_share_n1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
