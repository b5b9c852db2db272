#!/usr/bin/env python
"""tools/exp/nms_stats.py -- candidate counts, kept counts and the scan's exit position of the bench's NMS calls."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from probabilisticteacher_amd import ops  # noqa: E402

orig = ops.nms_batched


def hooked(boxes_sorted, seg_offsets, max_count, thr, max_keep, seg_counts=None):
    keep, cnt = orig(boxes_sorted, seg_offsets, max_count, thr, max_keep, seg_counts)
    so = seg_offsets.cpu()
    n = (so[1:] - so[:-1])
    if seg_counts is not None:
        n = torch.minimum(n, seg_counts.cpu())
    c = cnt.cpu()
    k = keep.cpu()
    last = torch.tensor([int(k[i, c[i] - 1]) if c[i] > 0 else -1 for i in range(len(c))])
    full = c >= max_keep
    print(f"nms imgs={len(c)} max_count={max_count} thr={thr} max_keep={max_keep} n mean={float(n.float().mean()):.0f} kept mean={float(c.float().mean()):.0f} "
          f"full={int(full.sum())} exit position (last kept index) mean={float(last.float().mean()):.0f} max={int(last.max())}", flush=True)
    return keep, cnt


ops.nms_batched = hooked
sys.argv = [sys.argv[0], "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--pmc-traffic", "off"] + sys.argv[1:]
bench.main()
