"""Two-crop mapper and aspect-ratio grouping with the pixel work on the device.

Reference: pt/data/dataset_mapper.py:88-172 `DatasetMapperTwoCropSeparate.__call__` -- weak augmentation (D2
ResizeShortestEdge + RandomFlip) -> `image_weak_aug`; strong augmentation of a copy -> the record pair
(strong, weak) in the a0 record format ("image" uint8 (3,H,W), "instances" FreeInstances{gt_boxes, gt_classes}, height,
width); pt/data/common.py:106-180 `AspectRatioGroupedSemiSupDatasetTwoCrop`.

Scope: decoding and resizing stay on the host side of the boundary (the mapper takes the decoded image at training
resolution); horizontal flip, the four strong augmentations and the batching run here, for a whole step's images at once."""
import random
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from ..structures import Boxes, FreeInstances
from .augment import StrongParams, hflip_batch, sample_strong_params, strong_augment_batch


class DeviceTwoCropMapper:
    """dataset dicts {"image": uint8 (3,H,W) tensor (any device), "boxes": (M,4) xyxy abs, "classes": (M,), "height",
    "width"} -> list of (strong record, weak record) pairs, as DatasetMapperTwoCropSeparate returns per image."""

    def __init__(self, device, flip_prob: float = 0.5, seed: Optional[int] = None, min_box_side: float = 1e-5):
        self.device = torch.device(device)
        self.flip_prob = flip_prob
        self.rng = random.Random(seed)
        self.min_box_side = min_box_side

    def __call__(self, dataset_dicts: Sequence[Dict], params: Optional[Sequence[StrongParams]] = None,
                 flips: Optional[Sequence[bool]] = None) -> List[Tuple[Dict, Dict]]:
        n = len(dataset_dicts)
        flips = list(flips) if flips is not None else [self.rng.random() < self.flip_prob for _ in range(n)]
        params = list(params) if params is not None else [sample_strong_params(self.rng) for _ in range(n)]
        imgs = [d["image"].to(self.device, non_blocking=True) for d in dataset_dicts]
        weak = hflip_batch(imgs, flips)
        strong = strong_augment_batch(weak, params)
        out = []
        for d, w_img, s_img, flip in zip(dataset_dicts, weak, strong, flips):
            h, w = w_img.shape[-2:]
            inst = None
            if "boxes" in d:
                b = d["boxes"].to(self.device).float().clone()
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


class AspectRatioGroupedSemiSupDatasetTwoCrop:
    """pt/data/common.py:106-180: two streams of (strong, weak) record pairs -> batches
    (label_strong, label_weak, unlabel_strong, unlabel_weak), images with w > h and w <= h kept in separate buckets."""

    def __init__(self, dataset: Tuple[Iterable, Iterable], batch_size: Tuple[int, int]):
        self.label_dataset, self.unlabel_dataset = dataset
        self.batch_size_label, self.batch_size_unlabel = batch_size
        self._label_buckets = [[] for _ in range(2)]
        self._label_buckets_key = [[] for _ in range(2)]
        self._unlabel_buckets = [[] for _ in range(2)]
        self._unlabel_buckets_key = [[] for _ in range(2)]

    def __iter__(self):
        label_bucket, unlabel_bucket = [], []
        label_key, unlabel_key = [], []
        for d_label, d_unlabel in zip(self.label_dataset, self.unlabel_dataset):
            if len(label_bucket) != self.batch_size_label:
                bid = 0 if d_label[0]["width"] > d_label[0]["height"] else 1
                label_bucket, label_key = self._label_buckets[bid], self._label_buckets_key[bid]
                label_bucket.append(d_label[0])
                label_key.append(d_label[1])
            if len(unlabel_bucket) != self.batch_size_unlabel:
                bid = 0 if d_unlabel[0]["width"] > d_unlabel[0]["height"] else 1
                unlabel_bucket, unlabel_key = self._unlabel_buckets[bid], self._unlabel_buckets_key[bid]
                unlabel_bucket.append(d_unlabel[0])
                unlabel_key.append(d_unlabel[1])
            if len(label_bucket) == self.batch_size_label and len(unlabel_bucket) == self.batch_size_unlabel:
                yield (label_bucket[:], label_key[:], unlabel_bucket[:], unlabel_key[:])
                del label_bucket[:], label_key[:], unlabel_bucket[:], unlabel_key[:]
