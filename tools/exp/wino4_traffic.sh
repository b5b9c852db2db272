#!/bin/bash
# HBM-side traffic of the F(4x4,3x3) forward kernel alone: tools/exp/wino4_traffic.sh <lib.so | ""> [layer] [n]
cd /tmp && export TMPDIR=/tmp
LIB=$1; LAYER=${2:-conv4_2}; N=${3:-48}; FAM=${4:-wino4}      # FAM: wino4 | wino4p (the kernel whose dispatches are summed)
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/w4_tr*
LIBARG=""; [ -n "$LIB" ] && LIBARG="--lib $R/$LIB"
PARG=""; [ "$FAM" = "wino4p" ] && PARG="--p"
for i in 0 1; do
  case $i in
    0) C="FETCH_SIZE";;
    1) C="WRITE_SIZE";;
  esac
  rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/w4_tr$i -o p --output-format csv -- python $R/tools/exp/wino4_bench.py --only4 --n $N --iters 1 --reps 1 --layers $LAYER $LIBARG $PARG > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$R/gpurun_out/w4_tr*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "${FAM}_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
f = agg["FETCH_SIZE"] / max(n["FETCH_SIZE"], 1) * 1024 * 2; w = agg["WRITE_SIZE"] / max(n["WRITE_SIZE"], 1) * 1024
print("${LIB:-product} ${FAM} $LAYER n=$N: fetch (x2 corrected) %.3f GB  write %.3f GB per dispatch (%d dispatches)" % (f / 1e9, w / 1e9, n["FETCH_SIZE"]))
PY
rm -rf $R/gpurun_out/w4_tr*
