#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
for V in epistamp epi_k epistamp epi_k; do echo "== $V"
for L in conv1_2 conv3_2; do
python tools/exp/wino4_bench.py --only4 --p --stamps --n 16 --layers $L --lib tools/exp/_bin/libptmi355_wino4p_$V.so 2>&1 | grep -v amdgpu.ids
done; done > $O/r06_wino4p_epi_k.txt 2>&1
cat $O/r06_wino4p_epi_k.txt | cut -c1-250
