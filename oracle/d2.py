"""oracle/d2.py -- TEST INFRASTRUCTURE (CPU oracle).

Restatement of the detectron2==0.5 / torchvision==0.8.2 / fvcore primitives that the
reference's hot path calls.  Those packages are pinned by the reference only in prose
(/root/reference/README.md:15, /root/reference/requirements.txt:5-6), are NOT vendored
under /root/reference and are not installable here  =>  **parity unpinned**: the
behaviour below follows the published algorithms as recorded in SURVEY.md Appendix A
and is guarded by brute-force/property tests (tests/test_oracle_d2.py).

Each item cites the reference call site that depends on it.
"""
from __future__ import annotations

import ctypes
import itertools
import math
import os
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

# --------------------------------------------------------------------------- #
# plain-C helpers (NMS, ROIAlign)
# --------------------------------------------------------------------------- #
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        from . import build as _b

        path = _b.build()
        lib = ctypes.CDLL(path)
        lib.ptref_nms.restype = ctypes.c_int64
        lib.ptref_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                  ctypes.c_float, ctypes.c_void_p]
        for f in (lib.ptref_roi_align_fwd, lib.ptref_roi_align_bwd):
            f.restype = None
            f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + \
                [ctypes.c_int] * 6 + [ctypes.c_float]
        lib.ptref_bilinear_shrink_u8.restype = None
        lib.ptref_bilinear_shrink_u8.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5
        _LIB = lib
    return _LIB


def bilinear_shrink_u8(img: torch.Tensor, dh: int, dw: int) -> torch.Tensor:
    """uint8 (C,H,W) -> uint8 (C,dh,dw): ATen's CPU bilinear (align_corners=False) + truncation, in the evaluation
    order of its multi-threaded generic kernel (ref_ops.c: ptref_bilinear_shrink_u8) -- independent of this host's
    torch thread count / CPU capability, unlike F.interpolate itself."""
    img = img.contiguous()
    assert img.dtype == torch.uint8 and img.dim() == 3
    out = torch.empty((img.shape[0], dh, dw), dtype=torch.uint8)
    _lib().ptref_bilinear_shrink_u8(img.data_ptr(), out.data_ptr(), img.shape[0], img.shape[1], img.shape[2], dh, dw)
    return out


def cat(tensors: List[torch.Tensor], dim: int = 0) -> torch.Tensor:
    """detectron2.layers.cat: torch.cat that skips the copy for a single tensor."""
    assert isinstance(tensors, (list, tuple))
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


def nonzero_tuple(x: torch.Tensor):
    if x.dim() == 0:
        return x.unsqueeze(0).nonzero().unbind(1)
    return x.nonzero().unbind(1)


def cross_entropy(input, target, *, reduction="mean", **kwargs):
    """detectron2.layers.cross_entropy: returns 0*sum for empty targets w/ mean."""
    if target.numel() == 0 and reduction == "mean":
        return input.sum() * 0.0
    return torch.nn.functional.cross_entropy(input, target, reduction=reduction, **kwargs)


# --------------------------------------------------------------------------- #
# A.1 Boxes  (used everywhere, e.g. proposal_utils.py:111,128,131)
# --------------------------------------------------------------------------- #
class Boxes:
    def __init__(self, tensor: torch.Tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        tensor = tensor.to(torch.float32) if tensor.dtype != torch.float32 else tensor
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self) -> "Boxes":
        return Boxes(self.tensor.clone())

    def to(self, *args, **kwargs) -> "Boxes":
        return Boxes(self.tensor.to(*args, **kwargs))

    def area(self) -> torch.Tensor:
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def scale(self, scale_x: float, scale_y: float) -> None:
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def clip(self, box_size: Tuple[int, int]) -> None:
        assert torch.isfinite(self.tensor).all(), "Box tensor contains infinite or NaN!"
        h, w = box_size
        x1 = self.tensor[:, 0].clamp(min=0, max=w)
        y1 = self.tensor[:, 1].clamp(min=0, max=h)
        x2 = self.tensor[:, 2].clamp(min=0, max=w)
        y2 = self.tensor[:, 3].clamp(min=0, max=h)
        self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

    def nonempty(self, threshold: float = 0.0) -> torch.Tensor:
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def inside_box(self, box_size, boundary_threshold: int = 0) -> torch.Tensor:
        h, w = box_size
        b = self.tensor
        return ((b[..., 0] >= -boundary_threshold) & (b[..., 1] >= -boundary_threshold)
                & (b[..., 2] < w + boundary_threshold) & (b[..., 3] < h + boundary_threshold))

    def __getitem__(self, item) -> "Boxes":
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, "Indexing on Boxes with {} failed".format(item)
        return Boxes(b)

    def __len__(self) -> int:
        return self.tensor.shape[0]

    def __repr__(self) -> str:
        return "Boxes(" + str(self.tensor) + ")"

    @classmethod
    def cat(cls, boxes_list: List["Boxes"]) -> "Boxes":
        assert isinstance(boxes_list, (list, tuple))
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    @property
    def device(self):
        return self.tensor.device

    def __iter__(self):
        yield from self.tensor


# --------------------------------------------------------------------------- #
# A.2 pairwise_iou  (rpn.py:414, roi_heads.py:207-213)
# --------------------------------------------------------------------------- #
def pairwise_intersection(b1: Boxes, b2: Boxes) -> torch.Tensor:
    a, b = b1.tensor, b2.tensor
    wh = torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])
    wh.clamp_(min=0)
    return wh.prod(dim=2)


def pairwise_iou(b1: Boxes, b2: Boxes) -> torch.Tensor:
    area1, area2 = b1.area(), b2.area()
    inter = pairwise_intersection(b1, b2)
    return torch.where(inter > 0, inter / (area1[:, None] + area2 - inter),
                       torch.zeros(1, dtype=inter.dtype, device=inter.device))


# --------------------------------------------------------------------------- #
# A.3 Matcher  (rpn.py:415, roi_heads.py:214)
# --------------------------------------------------------------------------- #
class Matcher:
    def __init__(self, thresholds: List[float], labels: List[int],
                 allow_low_quality_matches: bool = False):
        thresholds = list(thresholds[:])
        assert thresholds[0] > 0
        thresholds.insert(0, -float("inf"))
        thresholds.append(float("inf"))
        assert all(lo <= hi for lo, hi in zip(thresholds[:-1], thresholds[1:]))
        assert all(l in [-1, 0, 1] for l in labels)
        assert len(labels) == len(thresholds) - 1
        self.thresholds = thresholds
        self.labels = list(labels)
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix: torch.Tensor):
        m = match_quality_matrix
        assert m.dim() == 2
        if m.numel() == 0:
            default_matches = m.new_full((m.size(1),), 0, dtype=torch.int64)
            default_labels = m.new_full((m.size(1),), self.labels[0], dtype=torch.int8)
            return default_matches, default_labels
        assert torch.all(m >= 0)
        matched_vals, matches = m.max(dim=0)
        match_labels = matches.new_full(matches.size(), 1, dtype=torch.int8)
        for l, low, high in zip(self.labels, self.thresholds[:-1], self.thresholds[1:]):
            low_high = (matched_vals >= low) & (matched_vals < high)
            match_labels[low_high] = l
        if self.allow_low_quality_matches:
            highest_quality_foreach_gt, _ = m.max(dim=1)
            _, pred_inds_with_highest_quality = nonzero_tuple(
                m == highest_quality_foreach_gt[:, None])
            match_labels[pred_inds_with_highest_quality] = 1
        return matches, match_labels


# --------------------------------------------------------------------------- #
# A.4 subsample_labels  (rpn.py:433 via RPN._subsample_labels; roi _sample_proposals)
# --------------------------------------------------------------------------- #
PermFn = Callable[[int], torch.Tensor]


def _default_perm(n: int) -> torch.Tensor:
    return torch.randperm(n)


def subsample_labels(labels: torch.Tensor, num_samples: int, positive_fraction: float,
                     bg_label: int, perm_fn: Optional[PermFn] = None):
    """Two consecutive permutation draws: first for positives, then negatives.
    `perm_fn(n)` lets parity tests inject the permutations."""
    perm_fn = perm_fn or _default_perm
    positive = nonzero_tuple((labels != -1) & (labels != bg_label))[0]
    negative = nonzero_tuple(labels == bg_label)[0]
    num_pos = int(num_samples * positive_fraction)
    num_pos = min(positive.numel(), num_pos)
    num_neg = num_samples - num_pos
    num_neg = min(negative.numel(), num_neg)
    if hasattr(perm_fn, "for_candidates"):      # permutations derived from per-element random keys (KeyedPerm)
        perm1 = perm_fn.for_candidates(positive, labels.numel())[:num_pos]
        perm2 = perm_fn.for_candidates(negative, labels.numel())[:num_neg]
    else:
        perm1 = perm_fn(positive.numel())[:num_pos]
        perm2 = perm_fn(negative.numel())[:num_neg]
    return positive[perm1], negative[perm2]


# --------------------------------------------------------------------------- #
# A.8 nms / batched_nms  (proposal_utils.py:140, fast_rcnn.py:104)
# --------------------------------------------------------------------------- #
def descending_order(scores: torch.Tensor) -> torch.Tensor:
    """Stable descending argsort: ties broken by ascending original index.
    (torchvision's order under ties is unspecified; this is the build's policy,
    shared by the HIP path.)"""
    return torch.sort(scores, descending=True, stable=True)[1]


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = boxes.detach().to(torch.float32).contiguous()
    order = descending_order(scores.detach()).contiguous()
    keep = torch.empty(n, dtype=torch.int64)
    nk = _lib().ptref_nms(b.data_ptr(), order.data_ptr(), n, float(iou_threshold),
                          keep.data_ptr())
    return keep[:nk]


def nms_bruteforce(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """O(n^2) pure-python greedy NMS used only to cross-check the C routine."""
    order = np.argsort(-scores, kind="stable")
    keep, sup = [], np.zeros(len(boxes), bool)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for a, i in enumerate(order):
        if sup[i]:
            continue
        keep.append(i)
        for j in order[a + 1:]:
            if sup[j]:
                continue
            w = np.float32(min(boxes[i, 2], boxes[j, 2])) - np.float32(max(boxes[i, 0], boxes[j, 0]))
            h = np.float32(min(boxes[i, 3], boxes[j, 3])) - np.float32(max(boxes[i, 1], boxes[j, 1]))
            inter = np.float32(max(w, np.float32(0))) * np.float32(max(h, np.float32(0)))
            iou = inter / np.float32(np.float32(area[i] + area[j]) - inter)
            if iou > np.float32(thr):
                sup[j] = True
    return np.asarray(keep, dtype=np.int64)


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor,
                iou_threshold: float) -> torch.Tensor:
    """detectron2 batched_nms (<40000 boxes path) -> torchvision batched_nms offset
    trick: boxes + idxs*(max_coordinate+1), computed in fp32, then plain nms."""
    assert boxes.shape[-1] == 4
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    boxes = boxes.float()
    if len(boxes) < 40000:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        boxes_for_nms = boxes + offsets[:, None]
        return nms(boxes_for_nms, scores, iou_threshold)
    result_mask = scores.new_zeros(scores.size(), dtype=torch.bool)
    for id in torch.unique(idxs).cpu().tolist():
        mask = (idxs == id).nonzero().view(-1)
        keep = nms(boxes[mask], scores[mask], iou_threshold)
        result_mask[mask[keep]] = True
    keep = result_mask.nonzero().view(-1)
    keep = keep[scores[keep].argsort(descending=True)]
    return keep


# --------------------------------------------------------------------------- #
# A.9 ROIAlign (aligned=True, sampling_ratio=0)  (roi_heads.py:68-73,126)
# --------------------------------------------------------------------------- #
class _ROIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, P, scale):
        feat_c = feat.detach().contiguous().float()
        rois_c = rois.detach().contiguous().float()
        N, C, H, W = feat_c.shape
        R = rois_c.shape[0]
        out = torch.zeros((R, C, P, P), dtype=torch.float32)
        if R > 0:
            _lib().ptref_roi_align_fwd(feat_c.data_ptr(), rois_c.data_ptr(), out.data_ptr(),
                                       N, C, H, W, R, P, float(scale))
        ctx.save_for_backward(rois_c)
        ctx.shape = (N, C, H, W)
        ctx.P, ctx.scale = P, scale
        return out

    @staticmethod
    def backward(ctx, gout):
        (rois_c,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        g = torch.zeros((N, C, H, W), dtype=torch.float32)
        R = rois_c.shape[0]
        if R > 0:
            go = gout.contiguous().float()
            _lib().ptref_roi_align_bwd(go.data_ptr(), rois_c.data_ptr(), g.data_ptr(),
                                       N, C, H, W, R, ctx.P, float(ctx.scale))
        return g, None, None, None


def roi_align(feat: torch.Tensor, rois: torch.Tensor, output_size: int,
              spatial_scale: float) -> torch.Tensor:
    """rois (R,5) = [batch_index, x1, y1, x2, y2] in image coordinates."""
    return _ROIAlignFn.apply(feat, rois, int(output_size), float(spatial_scale))


def roi_align_python(feat: np.ndarray, rois: np.ndarray, P: int, scale: float) -> np.ndarray:
    """Direct transcription of SURVEY.md A.9 in numpy (slow; cross-check only)."""
    N, C, H, W = feat.shape
    out = np.zeros((len(rois), C, P, P), np.float32)
    f32 = np.float32
    for r, roi in enumerate(rois):
        b = int(roi[0])
        sw, sh = f32(roi[1] * f32(scale)) - f32(.5), f32(roi[2] * f32(scale)) - f32(.5)
        ew, eh = f32(roi[3] * f32(scale)) - f32(.5), f32(roi[4] * f32(scale)) - f32(.5)
        rw, rh = ew - sw, eh - sh
        bh, bw = rh / f32(P), rw / f32(P)
        gh, gw = int(math.ceil(rh / f32(P))), int(math.ceil(rw / f32(P)))
        count = max(gh * gw, 1)
        for ph in range(P):
            for pw in range(P):
                acc = np.zeros(C, np.float32)
                for iy in range(gh):
                    y = sh + f32(ph) * bh + f32(iy + .5) * bh / f32(gh)
                    for ix in range(gw):
                        x = sw + f32(pw) * bw + f32(ix + .5) * bw / f32(gw)
                        if y < -1 or y > H or x < -1 or x > W:
                            continue
                        yy, xx = max(y, f32(0)), max(x, f32(0))
                        yl, xl = int(yy), int(xx)
                        if yl >= H - 1:
                            yl = yh = H - 1
                            yy = f32(yl)
                        else:
                            yh = yl + 1
                        if xl >= W - 1:
                            xl = xh = W - 1
                            xx = f32(xl)
                        else:
                            xh = xl + 1
                        ly, lx = f32(yy - yl), f32(xx - xl)
                        hy, hx = f32(1) - ly, f32(1) - lx
                        acc += (hy * hx * feat[b, :, yl, xl] + hy * lx * feat[b, :, yl, xh]
                                + ly * hx * feat[b, :, yh, xl] + ly * lx * feat[b, :, yh, xh])
                out[r, :, ph, pw] = acc / f32(count)
    return out


def convert_boxes_to_pooler_format(box_lists: List[Boxes]) -> torch.Tensor:
    """detectron2 ROIPooler helper: (R,5) with the image index in column 0."""
    rows = []
    for i, b in enumerate(box_lists):
        t = b.tensor
        rows.append(torch.cat([torch.full((len(t), 1), float(i), dtype=t.dtype), t], dim=1))
    if not rows:
        return torch.zeros((0, 5), dtype=torch.float32)
    return torch.cat(rows, dim=0)


# --------------------------------------------------------------------------- #
# A.14 Instances  (pt/structures/instances.py:22 subclasses it)
# --------------------------------------------------------------------------- #
class Instances:
    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        self._image_size = image_size
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self) -> Tuple[int, int]:
        return self._image_size

    def __setattr__(self, name: str, val: Any) -> None:
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name: str) -> Any:
        if name == "_fields" or name not in self._fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return self._fields[name]

    def set(self, name: str, value: Any) -> None:
        data_len = len(value)
        if len(self._fields):
            assert len(self) == data_len, \
                "Adding a field of length {} to a Instances of length {}".format(data_len, len(self))
        self._fields[name] = value

    def has(self, name: str) -> bool:
        return name in self._fields

    def remove(self, name: str) -> None:
        del self._fields[name]

    def get(self, name: str) -> Any:
        return self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def to(self, *args: Any, **kwargs: Any) -> "Instances":
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(*args, **kwargs)
            ret.set(k, v)
        return ret

    def __getitem__(self, item) -> "Instances":
        if type(item) == int:
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self) -> int:
        for v in self._fields.values():
            return v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    @staticmethod
    def cat(instance_lists: List["Instances"]) -> "Instances":
        assert all(isinstance(i, Instances) for i in instance_lists)
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        image_size = instance_lists[0].image_size
        for i in instance_lists[1:]:
            assert i.image_size == image_size
        ret = Instances(image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = list(itertools.chain(*values))
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError("Unsupported type {} for concatenation".format(type(v0)))
            ret.set(k, values)
        return ret


# --------------------------------------------------------------------------- #
# A.13 ImageList  (rcnn.py:40,43)
# --------------------------------------------------------------------------- #
class ImageList:
    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self) -> int:
        return len(self.image_sizes)

    def __getitem__(self, idx) -> torch.Tensor:
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    def to(self, *args, **kwargs) -> "ImageList":
        return ImageList(self.tensor.to(*args, **kwargs), self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def from_tensors(tensors: List[torch.Tensor], size_divisibility: int = 0,
                     pad_value: float = 0.0) -> "ImageList":
        assert len(tensors) > 0
        image_sizes = [(im.shape[-2], im.shape[-1]) for im in tensors]
        max_h = max(s[0] for s in image_sizes)
        max_w = max(s[1] for s in image_sizes)
        if size_divisibility > 1:
            st = size_divisibility
            max_h = (max_h + (st - 1)) // st * st
            max_w = (max_w + (st - 1)) // st * st
        batch_shape = [len(tensors)] + list(tensors[0].shape[:-2]) + [max_h, max_w]
        batched = tensors[0].new_full(batch_shape, pad_value)
        for img, pad_img in zip(tensors, batched):
            pad_img[..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batched.contiguous(), image_sizes)


# --------------------------------------------------------------------------- #
# A.5 anchors  (Guassian-RCNN-VGG.yaml:10-12; anchor_generator.py:117 reuses the grid)
# --------------------------------------------------------------------------- #
def create_grid_offsets(size: Sequence[int], stride: int, offset: float, device=None):
    grid_height, grid_width = size
    shifts_x = torch.arange(offset * stride, grid_width * stride, step=stride,
                            dtype=torch.float32, device=device)
    shifts_y = torch.arange(offset * stride, grid_height * stride, step=stride,
                            dtype=torch.float32, device=device)
    shift_y, shift_x = torch.meshgrid(shifts_y, shifts_x, indexing="ij")
    return shift_x.reshape(-1), shift_y.reshape(-1)


def default_cell_anchors(sizes=(128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0)) -> torch.Tensor:
    anchors = []
    for size in sizes:
        area = size ** 2.0
        for ar in aspect_ratios:
            w = math.sqrt(area / ar)
            h = ar * w
            anchors.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(anchors, dtype=torch.float32)


def grid_anchors(cell_anchors: torch.Tensor, grid_size: Sequence[int], stride: int,
                 offset: float) -> torch.Tensor:
    shift_x, shift_y = create_grid_offsets(grid_size, stride, offset, cell_anchors.device)
    shifts = torch.stack((shift_x, shift_y, shift_x, shift_y), dim=1)
    return (shifts.view(-1, 1, 4) + cell_anchors.view(1, -1, 4)).reshape(-1, 4)


def broadcast_params(params, num_features: int, name: str):
    assert isinstance(params, (list, tuple)), f"{name} in anchor generator has to be a list!"
    assert len(params), f"{name} in anchor generator cannot be empty!"
    if not isinstance(params[0], (list, tuple)):
        return [params] * num_features
    if len(params) == 1:
        return list(params) * num_features
    assert len(params) == num_features
    return params


# --------------------------------------------------------------------------- #
# weight init (fvcore.nn.weight_init)  (vgg.py:63; A.10)
# --------------------------------------------------------------------------- #
def c2_msra_fill(module: torch.nn.Module) -> None:
    torch.nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        torch.nn.init.constant_(module.bias, 0)


def c2_xavier_fill(module: torch.nn.Module) -> None:
    torch.nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        torch.nn.init.constant_(module.bias, 0)


# --------------------------------------------------------------------------- #
# A.15 LR schedule / optimiser
# --------------------------------------------------------------------------- #
def warmup_factor_at_iter(method: str, it: int, warmup_iters: int, warmup_factor: float) -> float:
    """detectron2 0.5 `_get_warmup_factor_at_iter` (imported by the reference at pt/solver/lr_scheduler.py:19)."""
    if it >= warmup_iters:
        return 1.0
    if method == "constant":
        return warmup_factor
    if method == "linear":
        alpha = it / warmup_iters
        return warmup_factor * (1 - alpha) + alpha
    raise ValueError("Unknown warmup method: {}".format(method))


def warmup_multistep_lr(it: int, base_lr: float, steps: Sequence[int], gamma: float = 0.1,
                        warmup_factor: float = 1e-3, warmup_iters: int = 1000,
                        warmup_method: str = "linear") -> float:
    import bisect

    if it >= warmup_iters:
        f = 1.0
    elif warmup_method == "constant":
        f = warmup_factor
    else:
        alpha = it / warmup_iters
        f = warmup_factor * (1 - alpha) + alpha
    return base_lr * f * gamma ** bisect.bisect_right(list(steps), it)
