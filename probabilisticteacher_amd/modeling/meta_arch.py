"""Meta-architecture: three-branch Gaussian Faster R-CNN (reference pt/modeling/meta_arch/rcnn.py:30-92) and the
teacher/student holder (pt/modeling/meta_arch/ts_ensemble.py:20-29)."""
from typing import List

import torch
from torch import nn

from .. import ops
from ..registry import META_ARCH_REGISTRY
from ..structures import ImageList
from .backbone import build_backbone
from .roi_heads import build_roi_heads
from .rpn import build_proposal_generator


@META_ARCH_REGISTRY.register()
class GuassianGeneralizedRCNN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        self.proposal_generator = build_proposal_generator(cfg, self.backbone.output_shape())
        self.roi_heads = build_roi_heads(cfg, self.backbone.output_shape())
        self.input_format = cfg.INPUT.FORMAT
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1), False)
        self._mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self._std = [float(v) for v in cfg.MODEL.PIXEL_STD]

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs: List[dict]) -> ImageList:
        """D2 GeneralizedRCNN.preprocess_image: normalise, then zero-pad to the batch max."""
        imgs = [x["image"].to(self.device, non_blocking=True) for x in batched_inputs]
        sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in imgs]
        return ImageList(ops.preprocess_images(imgs, self._mean, self._std), sizes)

    def inference(self, batched_inputs):
        """Eval-mode path (rcnn.py:33-34 -> D2 GeneralizedRCNN.inference + detector_postprocess):
        test-time RPN (6000 -> 1000 proposals), ROI inference, boxes rescaled to each record's height/width."""
        assert not self.training
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        proposals, _ = self.proposal_generator(images, features, None)
        results, _ = self.roi_heads(images, features, proposals, None)
        out = []
        for res, rec, size in zip(results, batched_inputs, images.image_sizes):
            out.append({"instances": detector_postprocess(res, rec.get("height", size[0]), rec.get("width", size[1]))})
        return out

    @staticmethod
    def _padded_size(batched_inputs):
        return (max(int(x["image"].shape[-2]) for x in batched_inputs), max(int(x["image"].shape[-1]) for x in batched_inputs))

    def can_run_jointly(self, sup_inputs, unsup_inputs) -> bool:
        """Both batches pad to the same canvas (then the padded region, the feature-map size and the anchor grid of each
        branch are what they would be in separate passes)."""
        return (self.training and len(sup_inputs) > 0 and len(unsup_inputs) > 0 and
                self._padded_size(sup_inputs) == self._padded_size(unsup_inputs))

    def forward_joint(self, sup_inputs, unsup_inputs, danchor=True):
        """`model(sup, branch="supervised")` and `model(unsup, branch="unsupervised", danchor=...)` (trainer.py:341,
        353-355) with ONE backbone + RPN-head pass over the concatenated images: the convolutions are per-image, so
        each branch sees exactly the activations of its separate pass; what changes is launch granularity (48 images
        per conv / wgrad launch instead of 32 + 16: fewer, fuller launches) and the summation order inside the weight
        gradients.  Returns (losses_sup, losses_unsup)."""
        assert self.can_run_jointly(sup_inputs, unsup_inputs)
        ns = len(sup_inputs)
        images = self.preprocess_image(list(sup_inputs) + list(unsup_inputs))
        features = self.backbone(images.tensor)
        feats = [features[f] for f in self.proposal_generator.in_features]
        obj, deltas = self.proposal_generator.head_outputs(feats)
        all_props = self.proposal_generator.proposals_from_head(images, features, (obj, deltas))
        out = []
        for sl, inputs, branch, da in ((slice(0, ns), sup_inputs, "supervised", False),
                                       (slice(ns, None), unsup_inputs, "unsupervised", danchor)):
            img = ImageList(images.tensor[sl], images.image_sizes[sl])
            feat = {k: v[sl] for k, v in features.items()}
            head = ([o[sl] for o in obj], [d[sl] for d in deltas])
            gt = [x["instances"].to(self.device) for x in inputs]
            proposals, l_rpn = self.proposal_generator(img, feat, gt, branch=branch if branch == "unsupervised" else "",
                                                       danchor=da, head_out=head, proposals=all_props[sl])
            _, l_det = self.roi_heads(img, feat, proposals, gt, branch=branch)
            losses = {}
            losses.update(l_det)
            losses.update(l_rpn)
            out.append(losses)
        return out[0], out[1]

    def forward(self, batched_inputs, branch="supervised", danchor=False):
        if not self.training:
            return self.inference(batched_inputs)
        images = self.preprocess_image(batched_inputs)
        gt_instances = None
        if "instances" in batched_inputs[0]:
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        features = self.backbone(images.tensor)

        if branch == "supervised":
            proposals_rpn, proposal_losses = self.proposal_generator(images, features, gt_instances)
            _, detector_losses = self.roi_heads(images, features, proposals_rpn, gt_instances, branch=branch)
            losses = {}
            losses.update(detector_losses)
            losses.update(proposal_losses)
            return losses, [], [], None
        if branch == "unsup_data_weak":
            proposals_rpn, _ = self.proposal_generator(images, features, None, compute_loss=False)
            proposals_roih, roi_predictions = self.roi_heads(images, features, proposals_rpn, targets=None,
                                                             compute_loss=False, branch=branch)
            return {}, proposals_rpn, proposals_roih, roi_predictions
        if branch == "unsupervised":
            proposals_rpn, proposal_losses = self.proposal_generator(images, features, gt_instances, branch=branch,
                                                                     danchor=danchor)
            _, detector_losses = self.roi_heads(images, features, proposals_rpn, gt_instances, branch=branch)
            losses = {}
            losses.update(detector_losses)
            losses.update(proposal_losses)
            return losses, [], [], None
        raise ValueError(f"unknown branch {branch!r}")


def detector_postprocess(results, output_height: int, output_width: int):
    """D2 0.5 detector_postprocess: rescale the detections from the network input size to the requested output
    size, clip, drop empty boxes."""
    scale_x = output_width / results.image_size[1]
    scale_y = output_height / results.image_size[0]
    res = type(results)((int(output_height), int(output_width)), **results.get_fields())
    boxes = res.pred_boxes.clone() if res.has("pred_boxes") else res.proposal_boxes.clone()
    boxes.scale(scale_x, scale_y)
    boxes.clip(res.image_size)
    res.set("pred_boxes" if res.has("pred_boxes") else "proposal_boxes", boxes)
    return res[boxes.nonempty()]


class EnsembleTSModel(nn.Module):
    """Holder that fixes the checkpoint prefixes `modelTeacher.` / `modelStudent.` (ts_ensemble.py:20-29)."""

    def __init__(self, modelTeacher, modelStudent):
        super().__init__()
        if isinstance(modelTeacher, nn.parallel.DistributedDataParallel):
            modelTeacher = modelTeacher.module
        if isinstance(modelStudent, nn.parallel.DistributedDataParallel):
            modelStudent = modelStudent.module
        self.modelTeacher = modelTeacher
        self.modelStudent = modelStudent


def build_model(cfg):
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
