#!/bin/bash
# PMC counters of the Winograd kernel alone (two passes): tools/exp/wino_pmc.sh <lib.so> [layer] -> prints per-kernel sums
cd /tmp && export TMPDIR=/tmp
LIB=$1; LAYER=${2:-conv3_2}
R=$GRAFT_REPO_ROOT
for i in 0 1 2; do
  case $i in
    0) C="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS";;
    1) C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL";;
    2) C="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL";;
  esac
  rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/wino_pmc$i -o p --output-format csv -- python $R/tools/exp/wino_bench.py --iters 1 --layers $LAYER $R/$LIB > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$R/gpurun_out/wino_pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "wino_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k in sorted(agg): print(f"{k:32s} {agg[k]/max(n[k],1):16.0f}  (per dispatch, {n[k]} dispatches)")
PY
