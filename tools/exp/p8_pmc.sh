#!/bin/bash
# PMC counters of the P8 (bf16-storage) kernels alone, three passes:   tools/exp/p8_pmc.sh [layer] [which] [n]   (on the GPU box)
# prints per-kernel per-dispatch averages for kernels whose name contains "p8_"
cd /tmp && export TMPDIR=/tmp
LAYER=${1:-conv3_2}; WHICH=${2:-conv}; N=${3:-16}
R=$GRAFT_REPO_ROOT
for i in 0 1 2 3; do
  case $i in
    0) C="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS";;
    1) C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL";;
    2) C="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL";;
    3) C="FETCH_SIZE GRBM_GUI_ACTIVE";;
  esac
  rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/p8_pmc$i -o p --output-format csv -- python $R/tools/kbench_p8.py --iters 1 --n $N --which $WHICH --layers $LAYER > /dev/null 2>&1
done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/p8_pmc4 -o p --output-format csv -- python $R/tools/kbench_p8.py --iters 1 --n $N --which $WHICH --layers $LAYER > /dev/null 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$R/gpurun_out/p8_pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "p8_" in k and "from_nchw" not in k and "pack" not in k:
            k = k.split("(")[0][-40:]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
for k in agg:
    print("==", k)
    for c in sorted(agg[k]): print(f"  {c:32s} {agg[k][c]/max(n[k][c],1):16.0f}  (per dispatch, {n[k][c]} dispatches)")
PY
