"""Anchor generators (D2 DefaultAnchorGenerator; reference pt/modeling/anchor_generator.py:31-163).

Both produce the (h*w*A, 4) anchor grid with a HIP kernel; the differentiable variant keeps the 9x(w,h) anchor
table as a Parameter named `anchor_0` and back-propagates into it (a segmented sum over grid cells)."""
import math
from typing import List

import torch
from torch import nn

from .. import ops
from ..registry import ANCHOR_GENERATOR_REGISTRY
from ..structures import Boxes


def _broadcast_params(params, num_features, name):
    assert isinstance(params, (list, tuple)) and len(params), f"{name} in anchor generator has to be a non-empty list!"
    if not isinstance(params[0], (list, tuple)):
        return [params] * num_features
    if len(params) == 1:
        return list(params) * num_features
    assert len(params) == num_features
    return params


class _GridAnchors(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cell, h, w, stride, offset):
        ctx.a = cell.shape[0]
        return ops.grid_anchors(cell.detach(), h, w, stride, offset)

    @staticmethod
    def backward(ctx, g):
        return g.view(-1, ctx.a, 4).sum(0), None, None, None, None


class DefaultAnchorGenerator(nn.Module):
    box_dim = 4

    def __init__(self, cfg, input_shape):
        super().__init__()
        self.strides = [s.stride for s in input_shape]
        nf = len(self.strides)
        sizes = _broadcast_params(cfg.MODEL.ANCHOR_GENERATOR.SIZES, nf, "sizes")
        ratios = _broadcast_params(cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS, nf, "aspect_ratios")
        self.offset = cfg.MODEL.ANCHOR_GENERATOR.OFFSET
        assert 0.0 <= self.offset < 1.0
        cells = []
        for s, r in zip(sizes, ratios):
            rows = []
            for size in s:
                area = size ** 2.0
                for ar in r:
                    w = math.sqrt(area / ar)
                    h = ar * w
                    rows.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
            cells.append(torch.tensor(rows, dtype=torch.float32))
        for i, c in enumerate(cells):
            self.register_buffer(f"cell_anchors_{i}", c, persistent=False)
        self._n = len(cells)

    @property
    def num_anchors(self):
        return [len(getattr(self, f"cell_anchors_{i}")) for i in range(self._n)]

    def forward(self, features: List[torch.Tensor]) -> List[Boxes]:
        out = []
        for i, f in enumerate(features):
            cell = getattr(self, f"cell_anchors_{i}")
            out.append(Boxes(ops.grid_anchors(cell, f.shape[-2], f.shape[-1], float(self.strides[i]), self.offset)))
        return out


class DifferentiableAnchorGenerator(nn.Module):
    """Anchors regenerated every forward from the learnable table `anchor_0` (9 rows of (w,h))."""
    box_dim = 4

    def __init__(self, cfg, input_shape):
        super().__init__()
        self.strides = [s.stride for s in input_shape]
        nf = len(self.strides)
        tables = _broadcast_params(cfg.MODEL.ANCHOR_GENERATOR.ANCHOR, nf, "sizes")
        self.offset = cfg.MODEL.ANCHOR_GENERATOR.OFFSET
        assert 0.0 <= self.offset < 1.0
        for i, t in enumerate(tables):
            self.register_parameter(f"anchor_{i}", nn.Parameter(torch.tensor(t, dtype=torch.float32)))
        self._n = len(tables)

    @property
    def num_anchors(self):
        return [len(getattr(self, f"anchor_{i}")) for i in range(self._n)]

    def forward(self, features: List[torch.Tensor]) -> List[Boxes]:
        out = []
        for i, f in enumerate(features):
            a = getattr(self, f"anchor_{i}")
            cell = torch.stack([-a[:, 0] / 2.0, -a[:, 1] / 2.0, a[:, 0] / 2.0, a[:, 1] / 2.0], -1)
            out.append(Boxes(_GridAnchors.apply(cell, f.shape[-2], f.shape[-1], float(self.strides[i]), self.offset)))
        return out


ANCHOR_GENERATOR_REGISTRY.register(DefaultAnchorGenerator)
ANCHOR_GENERATOR_REGISTRY.register(DifferentiableAnchorGenerator)


def build_anchor_generator(cfg, input_shape):
    return ANCHOR_GENERATOR_REGISTRY.get(cfg.MODEL.ANCHOR_GENERATOR.NAME)(cfg, input_shape)
