#!/usr/bin/env python
"""tools/exp/copy_census.py <kernel_trace.csv> [pattern] -- what surrounds the launches matching `pattern` (default: the runtime's
copyBuffer blits) in the last step of a rocprofv3 kernel trace: (previous kernel, next kernel) pairs by count."""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
ev = sorted(((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in rows))
opt = [i for i, e in enumerate(ev) if "clip_sgd" in e[1]]
if len(opt) >= 2:
    ev = ev[opt[-2] + 1:opt[-1] + 1]
c = Counter()
for i, (t, name) in enumerate(ev):
    if pat in name:
        prev = next((ev[j][1] for j in range(i - 1, -1, -1) if pat not in ev[j][1]), "-")
        nxt = next((ev[j][1] for j in range(i + 1, len(ev)) if pat not in ev[j][1]), "-")
        c[(prev[:60], nxt[:60])] += 1
print(sum(c.values()), "launches of", pat)
for (a, b), n in c.most_common(25):
    print(f"{n:4d}  {a}  ->  {b}")
