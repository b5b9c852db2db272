#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
(time timeout 3400 python -m pytest tests -m gpu -x -q --durations=25) > $O/r06_gputests_full.txt 2>&1; tail -45 $O/r06_gputests_full.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.txt 2>&1; tail -3 $O/r06_smoke.txt | cut -c1-200
