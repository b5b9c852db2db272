#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
for R in 1 2; do
echo "== product (n=48)"; python tools/exp/wino4_bench.py --only4 --p --n 48 --iters 3 2>&1 | grep -v amdgpu.ids
echo "== vmcnt(0) inside the epilogue (n=48)"; python tools/exp/wino4_bench.py --only4 --p --n 48 --iters 3 --lib tools/exp/_bin/libptmi355_wino4p_vm.so 2>&1 | grep -v amdgpu.ids
done
echo "== product (n=16, epi 3)"; python tools/exp/wino4_bench.py --only4 --p --n 16 --epi 3 2>&1 | grep -v amdgpu.ids
echo "== variant (n=16, epi 3)"; python tools/exp/wino4_bench.py --only4 --p --n 16 --epi 3 --lib tools/exp/_bin/libptmi355_wino4p_vm.so 2>&1 | grep -v amdgpu.ids
echo "== stamps"; python tools/exp/wino4_bench.py --only4 --p --stamps --n 16 --layers conv1_2,conv3_2 --lib tools/exp/_bin/libptmi355_wino4p_vm_st.so 2>&1 | grep -v amdgpu.ids
} > $O/r06_wino4p_vmcnt_in_epilogue.txt 2>&1
cat $O/r06_wino4p_vmcnt_in_epilogue.txt | cut -c95-250
