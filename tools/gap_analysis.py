#!/usr/bin/env python
"""Where is the GPU idle?  Reads a rocprofv3 --kernel-trace CSV and reports, for the last step of the run, the idle time
between consecutive kernels grouped by the kernel that FOLLOWS the gap (i.e. what the GPU was waiting to be given)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# last step = from the last ema_kernel (first kernel of run_step) to the last clip_sgd_kernel (its last kernel)
idx = [i for i, e in enumerate(ev) if "ema_kernel" in e[2]]
end = [i for i, e in enumerate(ev) if "clip_sgd_kernel" in e[2]]
lo = idx[-1] if idx else 0
hi = end[-1] + 1 if end else len(ev)
ev = ev[lo:hi]
span = ev[-1][1] - ev[0][0]
busy = 0
gaps = collections.Counter()
ngaps = collections.Counter()
cur_end = ev[0][0]
for s, e, name in ev:
    if s > cur_end:
        g = s - cur_end
        key = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        gaps[key] += g
        ngaps[key] += 1
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
print(f"last step: span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, idle {(span - busy) / 1e6:.1f} ms, kernels {len(ev)}")
for k, v in gaps.most_common(25):
    print(f"  idle before {k:70s} {v / 1e6:7.2f} ms in {ngaps[k]:4d} gaps")
small = collections.Counter()
for s, e, name in ev:
    if e - s < 50_000:
        small[name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]] += e - s
print(f"kernels shorter than 50 us: {sum(small.values()) / 1e6:.1f} ms")
