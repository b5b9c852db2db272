"""Strong augmentation of the two-crop mapper ON THE DEVICE.

Reference: pt/data/detection_utils.py:38-60 `build_strong_augmentation`
    RandomApply([ColorJitter(0.4, 0.4, 0.4, 0.1)], p=0.8); RandomGrayscale(p=0.2);
    RandomApply([GaussianBlur([0.1, 2.0])], p=0.5); RandomApply([Solarize(threshold=0.5)], p=0.2)
run per image on a PIL copy inside DataLoader workers (pt/data/dataset_mapper.py:151-159).  At ~45 img/s per GPU x 2
strong crops that is ~100 PIL pipelines per second per GPU; here the whole batch of a step is augmented by a handful of
HBM-bound launches (csrc/augment.hip), byte-exact with Pillow for given parameters.

The random PARAMETERS are drawn on the host (`sample_strong_params`, same distributions as torchvision 0.8.2 /
augmentation_impl.py); the pixels never leave the device."""
import math
import random
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch

from .. import _lib
from ..ops import _chk, _image_desc, _ptr, _stream

OP_COPY, OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION, OP_HUE, OP_GRAY, OP_SOLARIZE = range(7)


@dataclass
class StrongParams:
    """one image's draw: `jitter` = the ColorJitter ops in application order [(op, factor), ...] (empty: not applied)"""
    jitter: List[Tuple[int, float]] = field(default_factory=list)
    gray: bool = False
    blur_sigma: Optional[float] = None
    solarize: Optional[int] = None          # threshold (round(0.5 * 256) = 128 in the reference) or None


def sample_strong_params(rng: random.Random) -> StrongParams:
    """Same distributions as the reference's pipeline (torchvision 0.8.2 ColorJitter.get_params: uniform factors in
    [max(0, 1 - v), 1 + v], hue in [-0.1, 0.1], shuffled order; RandomApply / RandomGrayscale coin flips;
    augmentation_impl.py:36 sigma ~ U(0.1, 2.0); :43 threshold = round(0.5 * 256))."""
    p = StrongParams()
    if rng.random() < 0.8:
        ops = [(OP_BRIGHTNESS, rng.uniform(0.6, 1.4)), (OP_CONTRAST, rng.uniform(0.6, 1.4)),
               (OP_SATURATION, rng.uniform(0.6, 1.4)), (OP_HUE, rng.uniform(-0.1, 0.1))]
        rng.shuffle(ops)
        p.jitter = ops
    p.gray = rng.random() < 0.2
    if rng.random() < 0.5:
        p.blur_sigma = rng.uniform(0.1, 2.0)
    if rng.random() < 0.2:
        p.solarize = round(0.5 * 256)
    return p


def _f32_bits(x: float) -> int:
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


def hue_shift(hue_factor: float) -> int:
    """`np.uint8(hue_factor * 255)` (torchvision adjust_hue): C float -> uint8 conversion, truncation then modulo 256"""
    return int(hue_factor * 255) & 0xFF


def box_blur_weights(sigma: float) -> Tuple[int, int, int]:
    """ImageFilter.GaussianBlur(radius=sigma) -> Pillow's ImagingGaussianBlur (3 box passes): integer box radius and the
    24-bit fixed-point weights ww (inner taps) / fw (the two fractional end taps), evaluated in fp32 / fp64 as BoxBlur.c does."""
    import numpy as np
    f32 = np.float32
    passes = 3
    sigma2 = f32(f32(f32(sigma) * f32(sigma)) / f32(passes))
    L = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(L) - 1.0) / 2.0))
    a = f32(f32(f32(2) * l + f32(1)) * f32(f32(l * f32(l + f32(1))) - f32(f32(3) * sigma2)))
    fr = f32(l + f32(a / f32(f32(6) * f32(sigma2 - f32(f32(l + f32(1)) * f32(l + f32(1)))))))
    radius = int(fr)
    ww = int(f32(1 << 24) / f32(f32(fr * f32(2)) + f32(1)))
    fw = ((1 << 24) - (radius * 2 + 1) * ww) // 2
    return radius, ww, fw


def _launch(name: str, rows, dev, max_elems: int, *extra):
    desc = _image_desc(rows, dev)
    _lib.call(name, _ptr(desc), len(rows), int(max_elems), *extra, _stream())


def strong_augment_batch(images: Sequence[torch.Tensor], params: Sequence[StrongParams]) -> List[torch.Tensor]:
    """images: uint8 (3, H, W) device tensors (channel order as in the record; PIL is told "RGB", dataset_mapper.py:155).
    Returns new tensors; the inputs are not modified.  Launches per BATCH: <= 4 colour rounds (+ a grey-level sum before a
    round in which some image applies contrast), grayscale, 6 box-blur passes, solarize."""
    n = len(images)
    if n == 0:
        return []
    dev = images[0].device
    cur = [_chk(im.contiguous(), torch.uint8, "image").clone() for im in images]        # colour ops run in place on the copy
    hw = [int(im.shape[-2] * im.shape[-1]) for im in cur]
    sums = torch.zeros(n, dtype=torch.int64, device=dev)

    def color_round(items):
        """items: [(image index, op, float factor, int parameter)]"""
        if not items:
            return
        if any(op == OP_CONTRAST for _, op, _, _ in items):
            rows = [[cur[i].data_ptr(), 0, cur[i].shape[-2], cur[i].shape[-1], 0, 0, 0, 0] for i, _, _, _ in items]
            desc = _image_desc(rows, dev)
            part = torch.empty(len(items), dtype=torch.int64, device=dev)
            _lib.call("ptmi_aug_gray_sum_batched", _ptr(desc), len(rows), max(hw[i] for i, _, _, _ in items), _ptr(part), _stream())
        else:
            part = sums
        rows = [[cur[i].data_ptr(), cur[i].data_ptr(), cur[i].shape[-2], cur[i].shape[-1], op, _f32_bits(f), ip, 0]
                for i, op, f, ip in items]
        _launch("ptmi_aug_color_batched", rows, dev, max(hw[i] for i, _, _, _ in items), _ptr(part))

    for k in range(4):
        items = []
        for i, p in enumerate(params):
            if len(p.jitter) > k:
                op, f = p.jitter[k]
                items.append((i, op, f if op != OP_HUE else 0.0, hue_shift(f) if op == OP_HUE else 0))
        color_round(items)
    color_round([(i, OP_GRAY, 0.0, 0) for i, p in enumerate(params) if p.gray])
    blur = [i for i, p in enumerate(params) if p.blur_sigma is not None]
    if blur:
        tmp = {i: torch.empty_like(cur[i]) for i in blur}
        dst = {i: torch.empty_like(cur[i]) for i in blur}
        wts = {i: box_blur_weights(params[i].blur_sigma) for i in blur}
        seq = [(cur, tmp), (tmp, dst), (dst, tmp), (tmp, dst), (dst, tmp), (tmp, dst)]       # H H H V V V, ends in dst
        for ps, (a, b) in enumerate(seq):
            rows = [[a[i].data_ptr(), b[i].data_ptr(), cur[i].shape[-2], cur[i].shape[-1], int(ps >= 3), *wts[i]] for i in blur]
            _launch("ptmi_aug_box_blur_batched", rows, dev, max(3 * hw[i] for i in blur))
        for i in blur:
            cur[i] = dst[i]
    color_round([(i, OP_SOLARIZE, 0.0, int(p.solarize)) for i, p in enumerate(params) if p.solarize is not None])
    return cur


def hflip_batch(images: Sequence[torch.Tensor], flips: Sequence[bool]) -> List[torch.Tensor]:
    """D2 RandomFlip(horizontal) -> HFlipTransform on planar uint8 images; one launch for the batch."""
    if not images:
        return []
    dev = images[0].device
    images = [_chk(im.contiguous(), torch.uint8, "image") for im in images]
    out = [torch.empty_like(im) for im in images]
    rows = [[im.data_ptr(), o.data_ptr(), im.shape[-2], im.shape[-1], int(bool(f)), 0, 0, 0] for im, o, f in zip(images, out, flips)]
    _launch("ptmi_aug_hflip_batched", rows, dev, max(im.numel() for im in images))
    return out


def resize_shortest_edge_size(h: int, w: int, short_edge: int, max_size: int) -> Tuple[int, int]:
    """D2 ResizeShortestEdge.get_transform: scale the short side to `short_edge`, cap the long side at `max_size`, round
    half up.  Returns (new_h, new_w)."""
    scale = short_edge * 1.0 / min(h, w)
    newh, neww = (short_edge, scale * w) if h < w else (scale * h, short_edge)
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def resize_batch(images: Sequence[torch.Tensor], sizes: Sequence[Tuple[int, int]]) -> List[torch.Tensor]:
    """Image.resize((new_w, new_h), Image.BILINEAR) of D2's ResizeTransform for a batch of planar uint8 images: the x pass
    of all images in one launch, then the y pass (a pass that does not change the size is skipped, as in Pillow)."""
    if not images:
        return []
    dev = images[0].device
    cur = [_chk(im.contiguous(), torch.uint8, "image") for im in images]
    for vertical in (0, 1):
        rows, outs, idx = [], [], []
        for i, (im, (nh, nw)) in enumerate(zip(cur, sizes)):
            h, w = im.shape[-2:]
            new = nh if vertical else nw
            if new == (h if vertical else w):
                continue
            if math.ceil(max((h if vertical else w) / new, 1.0)) * 2 + 1 > 32:
                raise ValueError(f"resize {h}x{w} -> {nh}x{nw}: down-scaling factor beyond the kernel's 32-tap window")
            o = torch.empty((3, new, w) if vertical else (3, h, new), dtype=torch.uint8, device=dev)
            rows.append([im.data_ptr(), o.data_ptr(), h, w, new, vertical, 0, 0])
            outs.append(o)
            idx.append(i)
        if rows:
            _launch("ptmi_aug_resize_pass_batched", rows, dev, max(o.numel() for o in outs))
            for i, o in zip(idx, outs):
                cur[i] = o
    return cur
