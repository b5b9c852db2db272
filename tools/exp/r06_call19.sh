#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
for L in conv1_2 conv3_2; do
python tools/exp/wino4_bench.py --only4 --p --stamps --n 16 --layers $L --lib tools/exp/_bin/libptmi355_wino4p_epistamp.so
python tools/exp/wino4_bench.py --only4 --p --stamps --n 16 --epi 3 --layers $L --lib tools/exp/_bin/libptmi355_wino4p_epistamp.so
done > $O/r06_wino4p_epi_stamps.txt 2>&1
cat $O/r06_wino4p_epi_stamps.txt | cut -c1-250
