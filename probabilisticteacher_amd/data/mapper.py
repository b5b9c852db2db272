"""Two-crop mapper and aspect-ratio grouping with the pixel work on the device.

Reference: pt/data/dataset_mapper.py:88-172 `DatasetMapperTwoCropSeparate.__call__` -- weak augmentation (D2
ResizeShortestEdge + RandomFlip) -> `image_weak_aug`; strong augmentation of a copy -> the record pair
(strong, weak) in the a0 record format ("image" uint8 (3,H,W), "instances" FreeInstances{gt_boxes, gt_classes}, height,
width); pt/data/common.py:106-180 `AspectRatioGroupedSemiSupDatasetTwoCrop`.

Scope: image decoding stays on the host side of the boundary (the mapper takes the decoded uint8 image); the weak
augmentation (ResizeShortestEdge = Pillow's antialiased bilinear resize, RandomFlip), the four strong augmentations and the
batching run here, for a whole step's images at once."""
import random
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from ..structures import Boxes, FreeInstances
from .augment import (StrongParams, hflip_batch, resize_batch, resize_shortest_edge_size, sample_strong_params,
                      strong_augment_batch)


class DeviceTwoCropMapper:
    """dataset dicts {"image": uint8 (3,H,W) tensor (any device), "boxes": (M,4) xyxy abs, "classes": (M,), "height",
    "width"} -> list of (strong record, weak record) pairs, as DatasetMapperTwoCropSeparate returns per image."""

    def __init__(self, device, flip_prob: float = 0.5, seed: Optional[int] = None, min_box_side: float = 1e-5,
                 min_size_train: Sequence[int] = (), max_size_train: int = 1333):
        """min_size_train / max_size_train = cfg.INPUT.MIN_SIZE_TRAIN / MAX_SIZE_TRAIN (sample_style "choice"); empty:
        the images already have their training resolution."""
        self.device = torch.device(device)
        self.flip_prob = flip_prob
        self.rng = random.Random(seed)
        self.min_box_side = min_box_side
        self.min_size_train, self.max_size_train = tuple(min_size_train), max_size_train

    @classmethod
    def from_config(cls, cfg, seed: Optional[int] = None):
        return cls(cfg.MODEL.DEVICE, flip_prob=0.5 if cfg.INPUT.RANDOM_FLIP == "horizontal" else 0.0, seed=seed,
                   min_size_train=cfg.INPUT.MIN_SIZE_TRAIN, max_size_train=cfg.INPUT.MAX_SIZE_TRAIN)

    def __call__(self, dataset_dicts: Sequence[Dict], params: Optional[Sequence[StrongParams]] = None,
                 flips: Optional[Sequence[bool]] = None, sizes: Optional[Sequence[Tuple[int, int]]] = None) -> List[Tuple[Dict, Dict]]:
        n = len(dataset_dicts)
        imgs = [d["image"].to(self.device, non_blocking=True) for d in dataset_dicts]
        if sizes is None:
            sizes = [resize_shortest_edge_size(im.shape[-2], im.shape[-1], self.rng.choice(self.min_size_train), self.max_size_train)
                     if self.min_size_train else tuple(im.shape[-2:]) for im in imgs]
        flips = list(flips) if flips is not None else [self.rng.random() < self.flip_prob for _ in range(n)]
        params = list(params) if params is not None else [sample_strong_params(self.rng) for _ in range(n)]
        weak = hflip_batch(resize_batch(imgs, sizes), flips)          # T.ResizeShortestEdge, then T.RandomFlip
        strong = strong_augment_batch(weak, params)
        out = []
        for d, src, w_img, s_img, flip in zip(dataset_dicts, imgs, weak, strong, flips):
            h, w = w_img.shape[-2:]
            inst = None
            if "boxes" in d:
                b = d["boxes"].to(self.device).float().clone()
                b[:, 0::2] *= w * 1.0 / src.shape[-1]        # ResizeTransform.apply_coords
                b[:, 1::2] *= h * 1.0 / src.shape[-2]
                if flip:                                     # HFlipTransform.apply_coords: x -> w - x, then re-order
                    x1 = w - b[:, 2]
                    x2 = w - b[:, 0]
                    b[:, 0], b[:, 2] = x1, x2
                b[:, 0::2].clamp_(0, w)                      # transform_instance_annotations clips to the image
                b[:, 1::2].clamp_(0, h)
                keep = ((b[:, 2] - b[:, 0]) > self.min_box_side) & ((b[:, 3] - b[:, 1]) > self.min_box_side)   # filter_empty_instances
                inst = FreeInstances((h, w))
                inst.gt_boxes = Boxes(b[keep])
                inst.gt_classes = d["classes"].to(self.device)[keep]
            base = {k: v for k, v in d.items() if k not in ("image", "boxes", "classes")}
            rec_s, rec_w = dict(base, image=s_img, height=h, width=w), dict(base, image=w_img, height=h, width=w)
            if inst is not None:
                rec_s["instances"], rec_w["instances"] = inst, inst      # the key record is a deepcopy in the reference
            out.append((rec_s, rec_w))
        return out


class _Stream:
    """one of the two record streams: two aspect-ratio buckets of (strong, weak) pairs and the bucket being filled"""

    def __init__(self, batch_size: int):
        self.batch_size = batch_size
        self.buckets = ([], [])            # index 0: landscape (w > h), 1: portrait / square
        self.current = self.buckets[0]

    def offer(self, pair) -> None:
        if len(self.current) == self.batch_size:       # a full batch is waiting for the other stream: drop the pair
            return
        strong = pair[0]
        self.current = self.buckets[0 if strong["width"] > strong["height"] else 1]
        self.current.append(pair)

    def ready(self) -> bool:
        return len(self.current) == self.batch_size

    def take(self):
        strong, weak = [p[0] for p in self.current], [p[1] for p in self.current]
        del self.current[:]
        return strong, weak


class AspectRatioGroupedSemiSupDatasetTwoCrop:
    """pt/data/common.py:106-180 (same name, same iteration contract): consumes a labelled and an unlabelled stream of
    (strong, weak) record pairs in lock step and yields (label_strong, label_weak, unlabel_strong, unlabel_weak) whenever
    BOTH streams have a full bucket; images with w > h and w <= h never share a batch (less padding).  As in the
    reference, a stream whose batch is already full ignores its incoming pairs until the other one catches up."""

    def __init__(self, dataset: Tuple[Iterable, Iterable], batch_size: Tuple[int, int]):
        self.label_dataset, self.unlabel_dataset = dataset
        self.batch_size_label, self.batch_size_unlabel = batch_size

    def __iter__(self):
        lab, unl = _Stream(self.batch_size_label), _Stream(self.batch_size_unlabel)
        for pair_l, pair_u in zip(self.label_dataset, self.unlabel_dataset):
            lab.offer(pair_l)
            unl.offer(pair_u)
            if lab.ready() and unl.ready():
                ls, lw = lab.take()
                us, uw = unl.take()
                yield ls, lw, us, uw
