"""The two-crop semi-supervised train loader and the test loader (reference pt/data/build.py:107-217, train_net.py:51-75).

Same structure as the reference -- an infinite, seeded, rank-sharded index stream per dataset (D2 TrainingSampler), the
two-crop mapper, aspect-ratio grouping of the labelled and the unlabelled stream in lock step, per-rank batch = total / world
(build.py:174-187) -- but the mapper is the device pipeline of data/mapper.py: the host only decodes the images."""
import itertools
from typing import Iterable, Iterator, List, Optional

import torch
import torch.distributed as dist

from ..structures import Boxes, FreeInstances
from . import datasets
from .augment import resize_batch, resize_shortest_edge_size
from .mapper import AspectRatioGroupedSemiSupDatasetTwoCrop, DeviceTwoCropMapper


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def training_sampler(size: int, seed: int = 0, shuffle: bool = True) -> Iterator[int]:
    """D2 TrainingSampler: an infinite stream of indices -- a fresh seeded permutation per epoch, identical on all ranks --
    of which rank r takes elements r, r + world, r + 2 world, ..."""
    rank, world = _rank_world()

    def stream():
        g = torch.Generator().manual_seed(seed)
        while True:
            yield from (torch.randperm(size, generator=g) if shuffle else torch.arange(size)).tolist()
    return itertools.islice(stream(), rank, None, world)


def _mapped_pairs(dicts: List[dict], sampler: Iterable[int], mapper: DeviceTwoCropMapper, fmt: str, labelled: bool):
    for i in sampler:
        d = datasets.to_mapper_input(dicts[i], fmt)
        if not labelled:                     # the unlabelled stream's annotations are never used (trainer.py:248-257)
            d = {k: v for k, v in d.items() if k not in ("boxes", "classes", "difficult")}
        yield mapper([d])[0]


def build_detection_semisup_train_loader_two_crops(cfg, mapper: Optional[DeviceTwoCropMapper] = None, seed: int = 0):
    """build.py:107-217: yields (label_strong, label_weak, unlabel_strong, unlabel_weak) lists of records forever"""
    rank, world = _rank_world()
    bl, bu = cfg.SOLVER.IMG_PER_BATCH_LABEL, cfg.SOLVER.IMG_PER_BATCH_UNLABEL
    assert bl > 0 and bl % world == 0, f"Total label batch size ({bl}) must be divisible by the number of gpus ({world})."
    assert bu > 0 and bu % world == 0, f"Total unlabel batch size ({bu}) must be divisible by the number of gpus ({world})."
    label_dicts = datasets.get_dataset_dicts(cfg.DATASETS.TRAIN_LABEL, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)   # build.py:111
    unlabel_dicts = datasets.get_dataset_dicts(cfg.DATASETS.TRAIN_UNLABEL, filter_empty=False)
    mapper = mapper or DeviceTwoCropMapper.from_config(cfg, seed=seed + 17 * rank)
    fmt = cfg.INPUT.FORMAT
    lab = _mapped_pairs(label_dicts, training_sampler(len(label_dicts), seed), mapper, fmt, True)
    unl = _mapped_pairs(unlabel_dicts, training_sampler(len(unlabel_dicts), seed + 1), mapper, fmt, False)
    return iter(AspectRatioGroupedSemiSupDatasetTwoCrop((lab, unl), (bl // world, bu // world)))


def build_detection_test_loader(cfg, dataset_name: str, batch_size: int = 1):
    """D2 build_detection_test_loader + DatasetMapper(is_train=False): ResizeShortestEdge(MIN_SIZE_TEST, MAX_SIZE_TEST), no flip;
    records keep the ORIGINAL height / width (detector_postprocess scales the detections back) and carry the ground truth in
    original coordinates for the evaluator.  Rank r evaluates images r, r + world, ... (D2 InferenceSampler shards contiguously;
    the evaluator gathers the shards, so the split does not matter)."""
    rank, world = _rank_world()
    dicts = datasets.get_dataset_dicts([dataset_name])
    dev = torch.device(cfg.MODEL.DEVICE)
    mine = dicts[rank::world]
    for s in range(0, len(mine), batch_size):
        batch = []
        for d in mine[s:s + batch_size]:
            m = datasets.to_mapper_input(d, cfg.INPUT.FORMAT)
            img = m["image"].to(dev)
            h, w = img.shape[-2:]
            nh, nw = resize_shortest_edge_size(h, w, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
            rec = {"image": resize_batch([img], [(nh, nw)])[0], "height": h, "width": w, "image_id": d["image_id"],
                   "file_name": d["file_name"]}
            if "boxes" in m:
                inst = FreeInstances((h, w))
                inst.gt_boxes, inst.gt_classes, inst.difficult = Boxes(m["boxes"]), m["classes"], m["difficult"]
                rec["instances"] = inst
            batch.append(rec)
        yield batch
