#!/usr/bin/env python
"""tools/exp/gaps.py <kernel_trace.csv> -- GPU idle time between consecutive kernels of a rocprofv3 --kernel-trace run: total,
and the largest gaps with the kernels either side (where the host does not keep the queue fed)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# one step: from the end of the last-but-one optimiser launch (clip_sgd closes a step) to the end of the last one
opt = [e for e in ev if "clip_sgd" in e[2]]
if len(opt) >= 2:
    ev = [e for e in ev if opt[-2][1] <= e[0] and e[1] <= opt[-1][1]]
busy = sum(e[1] - e[0] for e in ev)
span = ev[-1][1] - ev[0][0]
gaps = defaultdict(lambda: [0, 0])
tot = 0
end = ev[0][1]
for i in range(1, len(ev)):
    g = ev[i][0] - end
    if g > 0:
        tot += g
        k = (ev[i - 1][2][:50], ev[i][2][:50])
        gaps[k][0] += g
        gaps[k][1] += 1
    end = max(end, ev[i][1])
print(f"last step of the trace: span {span / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, idle {tot / 1e6:.2f} ms ({100 * tot / span:.1f} %), {len(ev)} launches")
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{g / 1e3:9.1f} us in {n:4d} gaps   {k[0]}  ->  {k[1]}")
