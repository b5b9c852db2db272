"""R-CNN box codec on HIP kernels (reference pt/modeling/box_regression.py:43-139)."""
import math
from typing import Tuple

import torch

from .. import ops

_DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)


class Box2BoxTransform:
    def __init__(self, weights: Tuple[float, float, float, float], scale_clamp: float = _DEFAULT_SCALE_CLAMP):
        self.weights = tuple(float(w) for w in weights)
        self.scale_clamp = scale_clamp

    def get_deltas(self, src_boxes: torch.Tensor, target_boxes: torch.Tensor) -> torch.Tensor:
        """(dx,dy,dw,dh) with the reference's `+1e-9` inside the logs (box_regression.py:94-95).  Differentiable
        w.r.t. src_boxes (the learnable anchors)."""
        return ops.get_deltas(src_boxes, target_boxes, self.weights)

    def apply_deltas(self, deltas: torch.Tensor, boxes: torch.Tensor) -> torch.Tensor:
        """deltas (N, 4k) -> boxes (N, 4k); dw/dh clamped to log(1000/16) (box_regression.py:101-139)."""
        return ops.apply_deltas(deltas.contiguous(), boxes, self.weights, self.scale_clamp)
