#!/usr/bin/env python
"""tools/exp/roi_stats.py -- extents (in feature cells) of the ROIs the bench's step hands to ROIAlign."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from probabilisticteacher_amd import ops  # noqa: E402

orig = ops.roi_align


def hooked(feat, rois, pooled, scale, img_offsets=None):
    r = rois.detach()
    w = ((r[:, 3] - r[:, 1]) * scale).cpu()
    h = ((r[:, 4] - r[:, 2]) * scale).cpu()
    q = torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9, 0.99])
    print(f"roi_align R={r.shape[0]} feat={tuple(feat.shape)} grad={feat.requires_grad} w cells q={[round(float(v), 1) for v in torch.quantile(w, q)]} "
          f"h cells q={[round(float(v), 1) for v in torch.quantile(h, q)]} mean area={float((w * h).mean()):.1f} "
          f"mean g={float((torch.ceil(w / 7).clamp(min=1) * torch.ceil(h / 7).clamp(min=1)).mean()):.2f}", flush=True)
    return orig(feat, rois, pooled, scale, img_offsets)


ops.roi_align = hooked
sys.argv = [sys.argv[0], "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--pmc-traffic", "off"]
bench.main()
