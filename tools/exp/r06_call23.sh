#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
for R in 1 2; do
echo "== product (n=48)"; python tools/exp/wino4_bench.py --only4 --p --n 48 --iters 3 2>&1 | grep -v amdgpu.ids
echo "== first-chunk C=0 (n=48)"; python tools/exp/wino4_bench.py --only4 --p --n 48 --iters 3 --lib tools/exp/_bin/libptmi355_wino4p_first.so 2>&1 | grep -v amdgpu.ids
done
echo "== product (n=16)"; python tools/exp/wino4_bench.py --only4 --p --n 16 2>&1 | grep -v amdgpu.ids
echo "== first-chunk C=0 (n=16)"; python tools/exp/wino4_bench.py --only4 --p --n 16 --lib tools/exp/_bin/libptmi355_wino4p_first.so 2>&1 | grep -v amdgpu.ids
echo "== first-chunk C=0, dgrad epilogue 3 (n=16)"; python tools/exp/wino4_bench.py --only4 --p --n 16 --epi 3 --lib tools/exp/_bin/libptmi355_wino4p_first.so 2>&1 | grep -v amdgpu.ids
echo "== stamps"; python tools/exp/wino4_bench.py --only4 --p --stamps --n 16 --layers conv1_2,conv3_2 --lib tools/exp/_bin/libptmi355_wino4p_first_st.so 2>&1 | grep -v amdgpu.ids
} > $O/r06_wino4p_first_chunk.txt 2>&1
cat $O/r06_wino4p_first_chunk.txt | cut -c95-250
