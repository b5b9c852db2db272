#!/usr/bin/env python
"""tools/exp/gap_list.py <kernel_trace.csv> [n] -- the n largest idle gaps of the trace's last step, each with the three kernels before
and after it (time offsets from the step's start in ms)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
opt = [e for e in ev if "clip_sgd" in e[2]]
if len(opt) >= 2:
    ev = [e for e in ev if opt[-2][1] <= e[0] and e[1] <= opt[-1][1]]
t0 = ev[0][0]
gaps = []
end = ev[0][1]
for i in range(1, len(ev)):
    g = ev[i][0] - end
    if g > 0:
        gaps.append((g, i))
    end = max(end, ev[i][1])
def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
    return name[:90]
for g, i in sorted(gaps, reverse=True)[:n]:
    print(f"gap {g / 1e3:8.1f} us at {(ev[i][0] - t0) / 1e6:8.2f} ms")
    for j in range(max(0, i - 3), min(len(ev), i + 3)):
        print(f"     {'>>' if j == i else '  '} {(ev[j][1] - ev[j][0]) / 1e3:8.1f} us  {short(ev[j][2])}")
