#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q --durations=5) > $O/r06g_gputests_full.txt 2>&1; tail -12 $O/r06g_gputests_full.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06g_smoke.txt 2>&1; tail -1 $O/r06g_smoke.txt | cut -c1-200
timeout 900 python bench.py > $O/r06g_bench_default.json 2> $O/r06g_bench_default.err; tail -c 1200 $O/r06g_bench_default.json
