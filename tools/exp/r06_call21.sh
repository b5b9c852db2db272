#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
for V in epistamp epi_nostore epi_nostore_nobar epi_nobar; do echo "== $V"
for L in conv1_2; do
python tools/exp/wino4_bench.py --only4 --p --stamps --n 16 --layers $L --lib tools/exp/_bin/libptmi355_wino4p_$V.so 2>&1 | grep -v amdgpu.ids
done; done > $O/r06_wino4p_epi_elimination.txt 2>&1
cat $O/r06_wino4p_epi_elimination.txt | cut -c1-250
