#!/bin/bash
# effective shader clock of the Winograd kernel: GRBM_GUI_ACTIVE / kernel duration (rocprofv3 kernel trace + one PMC pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for LIB in "$@"; do
rm -rf $R/gpurun_out/wino_clk
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $R/gpurun_out/wino_clk -o p --output-format csv -- python $R/tools/exp/wino_bench.py --iters 2 --layers conv3_2 $R/$LIB > /dev/null 2>&1
python - <<PY
import csv, glob, collections
d = "$R/gpurun_out/wino_clk"
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "wino" in row["Kernel_Name"] and "pack" not in row["Kernel_Name"]:
            dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
cnt = collections.defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Dispatch_Id"] in dur:
            cnt[row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
for k in sorted(dur, key=int):
    c = cnt[k]
    g = c.get("GRBM_GUI_ACTIVE", 0)
    print("$LIB", f"dispatch {k}: {dur[k]:9.1f} us  GRBM_GUI_ACTIVE {g:12.0f}  -> {g / dur[k] / 1e3:5.2f} GHz | wave_cyc(quad) {c.get('SQ_WAVE_CYCLES',0):.3e} wait_any {c.get('SQ_WAIT_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1):.3f} wait_inst {c.get('SQ_WAIT_INST_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1):.3f} active {c.get('SQ_ACTIVE_INST_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1):.3f} wait_lds {c.get('SQ_WAIT_INST_LDS',0)/max(c.get('SQ_WAVE_CYCLES',1),1):.3f}")
PY
done
