#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
for R in 1 2; do
echo "== product (n=48)"; python tools/exp/wino4_bench.py --only4 --p --n 48 --iters 3 2>&1 | grep -v amdgpu.ids
echo "== z-exchange (n=48)"; python tools/exp/wino4_bench.py --only4 --p --n 48 --iters 3 --lib tools/exp/_bin/libptmi355_wino4p_zx.so 2>&1 | grep -v amdgpu.ids
done
for E in 0 2 3; do
echo "== product (n=16, epi $E)"; python tools/exp/wino4_bench.py --only4 --p --n 16 --epi $E --layers conv1_2,conv3_2,conv5_1 2>&1 | grep -v amdgpu.ids
echo "== z-exchange (n=16, epi $E)"; python tools/exp/wino4_bench.py --only4 --p --n 16 --epi $E --layers conv1_2,conv3_2,conv5_1 --lib tools/exp/_bin/libptmi355_wino4p_zx.so 2>&1 | grep -v amdgpu.ids
done
echo "== stamps"; python tools/exp/wino4_bench.py --only4 --p --stamps --n 16 --layers conv1_2,conv3_2 --lib tools/exp/_bin/libptmi355_wino4p_zx_st.so 2>&1 | grep -v amdgpu.ids
} > $O/r06_wino4p_z_exchange.txt 2>&1
grep "==\|tile 1" $O/r06_wino4p_z_exchange.txt | cut -c1-200
awk '/== product \(n=48\)/{m="P"} /== z-exchange \(n=48\)/{m="Z"} /== product \(n=16/{m=""} /wino4p\/wino4/{ if(m!=""){split($0,a,"wino4p/wino4 x"); split(a[2],b," "); print m, $1, b[1], $NF} }' $O/r06_wino4p_z_exchange.txt | paste - - - - - - - -
grep -A3 "epi [023])" $O/r06_wino4p_z_exchange.txt | grep "conv\|==" | cut -c1-40,95-220
