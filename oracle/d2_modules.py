"""oracle/d2_modules.py -- TEST INFRASTRUCTURE (dev-container golden generation only).

Class-level stand-ins for the detectron2==0.5 framework pieces that the reference's
modules subclass (RPN, StandardRPNHead, StandardROIHeads, FastRCNNOutputLayers,
FastRCNNConvFCHead, ROIPooler, GeneralizedRCNN, Backbone, registries, `configurable`,
CfgNode).  detectron2 itself is absent from this image and cannot be installed, so
``tools/gen_golden.py`` mounts THESE under the ``detectron2.*`` names, imports the real
``/root/reference/pt`` modules on top, and records golden vectors.  Behaviour follows
SURVEY.md Appendix A ("parity unpinned" for everything in this file).

Never imported by the product package and never needed on the GPU box.
"""
from __future__ import annotations

import functools
import inspect
import math
from collections import namedtuple
from typing import Dict, List, Optional

import torch
from torch import nn
import torch.nn.functional as F

from . import d2
from .d2 import Boxes, ImageList, Instances, Matcher, cat, pairwise_iou

# injected by tools/gen_golden.py so that sampler draws are recorded / replayable
PERM_FN = None


# ----------------------------------------------------------------------------- config
class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def merge_from_dict(self, other: dict):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k].merge_from_dict(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else v

    def freeze(self):
        pass


def get_cfg() -> CfgNode:
    """Only the D2 0.5 defaults that the hot path reads."""
    C = CfgNode()
    C.VERSION = 2
    C.VIS_PERIOD = 0
    C.INPUT = CfgNode(dict(FORMAT="BGR", MIN_SIZE_TRAIN=(800,), MAX_SIZE_TRAIN=1333))
    C.SOLVER = CfgNode(dict(BASE_LR=0.001, MOMENTUM=0.9, WEIGHT_DECAY=1e-4, GAMMA=0.1, STEPS=(30000,),
                            WARMUP_FACTOR=1e-3, WARMUP_ITERS=1000, WARMUP_METHOD="linear", MAX_ITER=40000,
                            AMP=dict(ENABLED=False)))
    C.DATASETS = CfgNode(dict(TRAIN=(), TEST=()))
    C.TEST = CfgNode(dict(DETECTIONS_PER_IMAGE=100, EVAL_PERIOD=0))
    C.MODEL = CfgNode(dict(
        DEVICE="cpu", META_ARCHITECTURE="GeneralizedRCNN", MASK_ON=False, KEYPOINT_ON=False, LOAD_PROPOSALS=False,
        PIXEL_MEAN=[103.530, 116.280, 123.675], PIXEL_STD=[1.0, 1.0, 1.0], WEIGHTS="",
        BACKBONE=dict(NAME="build_resnet_backbone", FREEZE_AT=2),
        ANCHOR_GENERATOR=dict(NAME="DefaultAnchorGenerator", SIZES=[[32, 64, 128, 256, 512]],
                              ASPECT_RATIOS=[[0.5, 1.0, 2.0]], ANGLES=[[-90, 0, 90]], OFFSET=0.0),
        PROPOSAL_GENERATOR=dict(NAME="RPN", MIN_SIZE=0),
        RPN=dict(HEAD_NAME="StandardRPNHead", IN_FEATURES=["res4"], BOUNDARY_THRESH=-1, IOU_THRESHOLDS=[0.3, 0.7],
                 IOU_LABELS=[0, -1, 1], BATCH_SIZE_PER_IMAGE=256, POSITIVE_FRACTION=0.5, BBOX_REG_LOSS_TYPE="smooth_l1",
                 BBOX_REG_LOSS_WEIGHT=1.0, BBOX_REG_WEIGHTS=(1.0, 1.0, 1.0, 1.0), SMOOTH_L1_BETA=0.0, LOSS_WEIGHT=1.0,
                 PRE_NMS_TOPK_TRAIN=12000, PRE_NMS_TOPK_TEST=6000, POST_NMS_TOPK_TRAIN=2000, POST_NMS_TOPK_TEST=1000,
                 NMS_THRESH=0.7),
        ROI_HEADS=dict(NAME="Res5ROIHeads", NUM_CLASSES=80, IN_FEATURES=["res4"], IOU_THRESHOLDS=[0.5], IOU_LABELS=[0, 1],
                       BATCH_SIZE_PER_IMAGE=512, POSITIVE_FRACTION=0.25, SCORE_THRESH_TEST=0.05, NMS_THRESH_TEST=0.5,
                       PROPOSAL_APPEND_GT=True),
        ROI_BOX_HEAD=dict(NAME="", BBOX_REG_LOSS_TYPE="smooth_l1", BBOX_REG_LOSS_WEIGHT=1.0,
                          BBOX_REG_WEIGHTS=(10.0, 10.0, 5.0, 5.0), SMOOTH_L1_BETA=0.0, POOLER_RESOLUTION=14,
                          POOLER_SAMPLING_RATIO=0, POOLER_TYPE="ROIAlignV2", NUM_FC=0, FC_DIM=1024, NUM_CONV=0,
                          CONV_DIM=256, NORM="", CLS_AGNOSTIC_BBOX_REG=False, TRAIN_ON_PRED_BOXES=False),
    ))
    return C


def _called_with_cfg(*args, **kwargs) -> bool:
    if len(args) and isinstance(args[0], CfgNode):
        return True
    return isinstance(kwargs.get("cfg", None), CfgNode)


def configurable(init_func=None, *, from_config=None):
    assert init_func is not None and inspect.isfunction(init_func) and init_func.__name__ == "__init__"

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        if _called_with_cfg(*args, **kwargs):
            explicit = type(self).from_config(*args, **kwargs)
            init_func(self, **explicit)
        else:
            init_func(self, *args, **kwargs)

    return wrapped


class Registry:
    def __init__(self, name):
        self._name, self._map = name, {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._map[o.__name__] = o
                return o
            return deco
        self._map[obj.__name__] = obj

    def get(self, name):
        if name not in self._map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self._map[name]


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
RPN_HEAD_REGISTRY = Registry("RPN_HEAD")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")
ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")


class ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    def __new__(cls, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)


# ----------------------------------------------------------------------------- layers
class Conv2d(nn.Conv2d):
    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm, self.activation = norm, activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def get_norm(norm, out_channels):
    if norm is None or norm == "" or norm == "None":
        return None
    raise NotImplementedError(norm)


class CNNBlockBase(nn.Module):
    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        return self


class Backbone(nn.Module):
    @property
    def size_divisibility(self) -> int:
        return 0

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}


class _NullStorage:
    iter = 0

    def put_scalar(self, *a, **k):
        pass

    def put_scalars(self, *a, **k):
        pass


def get_event_storage():
    return _NullStorage()


def retry_if_cuda_oom(func):
    return func


# ----------------------------------------------------------------------------- anchors
class DefaultAnchorGenerator(nn.Module):
    box_dim = 4

    @configurable
    def __init__(self, *, sizes, aspect_ratios, strides, offset=0.5):
        super().__init__()
        self.strides = strides
        self.num_features = len(strides)
        sizes = d2.broadcast_params(sizes, self.num_features, "sizes")
        aspect_ratios = d2.broadcast_params(aspect_ratios, self.num_features, "aspect_ratios")
        self.cell_anchors = [d2.default_cell_anchors(s, a) for s, a in zip(sizes, aspect_ratios)]
        self.offset = offset

    @classmethod
    def from_config(cls, cfg, input_shape):
        return {"sizes": cfg.MODEL.ANCHOR_GENERATOR.SIZES, "aspect_ratios": cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS,
                "strides": [x.stride for x in input_shape], "offset": cfg.MODEL.ANCHOR_GENERATOR.OFFSET}

    @property
    def num_anchors(self):
        return [len(c) for c in self.cell_anchors]

    num_cell_anchors = num_anchors

    def forward(self, features):
        grid_sizes = [f.shape[-2:] for f in features]
        return [Boxes(d2.grid_anchors(c, g, s, self.offset))
                for g, s, c in zip(grid_sizes, self.strides, self.cell_anchors)]


ANCHOR_GENERATOR_REGISTRY.register(DefaultAnchorGenerator)


def build_anchor_generator(cfg, input_shape):
    return ANCHOR_GENERATOR_REGISTRY.get(cfg.MODEL.ANCHOR_GENERATOR.NAME)(cfg, input_shape)


# ----------------------------------------------------------------------------- RPN
class StandardRPNHead(nn.Module):
    @configurable
    def __init__(self, *, in_channels: int, num_anchors: int, box_dim: int = 4):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.objectness_logits = nn.Conv2d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.anchor_deltas = nn.Conv2d(in_channels, num_anchors * box_dim, kernel_size=1, stride=1)
        for l in [self.conv, self.objectness_logits, self.anchor_deltas]:
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)

    @classmethod
    def from_config(cls, cfg, input_shape):
        in_channels = [s.channels for s in input_shape]
        assert len(set(in_channels)) == 1
        ag = build_anchor_generator(cfg, input_shape)
        assert len(set(ag.num_anchors)) == 1
        return {"in_channels": in_channels[0], "num_anchors": ag.num_anchors[0], "box_dim": ag.box_dim}

    def forward(self, features: List[torch.Tensor]):
        obj, deltas = [], []
        for x in features:
            t = F.relu(self.conv(x))
            obj.append(self.objectness_logits(t))
            deltas.append(self.anchor_deltas(t))
        return obj, deltas


RPN_HEAD_REGISTRY.register(StandardRPNHead)


def build_rpn_head(cfg, input_shape):
    return RPN_HEAD_REGISTRY.get(cfg.MODEL.RPN.HEAD_NAME)(cfg, input_shape)


class RPN(nn.Module):
    @configurable
    def __init__(self, *, in_features, head, anchor_generator, anchor_matcher, box2box_transform,
                 batch_size_per_image, positive_fraction, pre_nms_topk, post_nms_topk, nms_thresh=0.7,
                 min_box_size=0.0, anchor_boundary_thresh=-1.0, loss_weight=1.0, box_reg_loss_type="smooth_l1",
                 smooth_l1_beta=0.0):
        super().__init__()
        self.in_features = in_features
        self.rpn_head = head
        self.anchor_generator = anchor_generator
        self.anchor_matcher = anchor_matcher
        self.box2box_transform = box2box_transform
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.pre_nms_topk = {True: pre_nms_topk[0], False: pre_nms_topk[1]}
        self.post_nms_topk = {True: post_nms_topk[0], False: post_nms_topk[1]}
        self.nms_thresh = nms_thresh
        self.min_box_size = float(min_box_size)
        self.anchor_boundary_thresh = anchor_boundary_thresh
        if isinstance(loss_weight, float):
            loss_weight = {"loss_rpn_cls": loss_weight, "loss_rpn_loc": loss_weight}
        self.loss_weight = loss_weight
        self.box_reg_loss_type = box_reg_loss_type
        self.smooth_l1_beta = smooth_l1_beta

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        in_features = cfg.MODEL.RPN.IN_FEATURES
        R = cfg.MODEL.RPN
        ret = {
            "in_features": in_features, "min_box_size": cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE,
            "nms_thresh": R.NMS_THRESH, "batch_size_per_image": R.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": R.POSITIVE_FRACTION,
            "loss_weight": {"loss_rpn_cls": R.LOSS_WEIGHT, "loss_rpn_loc": R.BBOX_REG_LOSS_WEIGHT * R.LOSS_WEIGHT},
            "anchor_boundary_thresh": R.BOUNDARY_THRESH, "box2box_transform": None,
            "box_reg_loss_type": R.BBOX_REG_LOSS_TYPE, "smooth_l1_beta": R.SMOOTH_L1_BETA,
            "pre_nms_topk": (R.PRE_NMS_TOPK_TRAIN, R.PRE_NMS_TOPK_TEST),
            "post_nms_topk": (R.POST_NMS_TOPK_TRAIN, R.POST_NMS_TOPK_TEST),
        }
        shapes = [input_shape[f] for f in in_features]
        ret["anchor_generator"] = build_anchor_generator(cfg, shapes)
        ret["anchor_matcher"] = Matcher(R.IOU_THRESHOLDS, R.IOU_LABELS, allow_low_quality_matches=True)
        ret["head"] = build_rpn_head(cfg, shapes)
        return ret

    def _subsample_labels(self, label):
        pos_idx, neg_idx = d2.subsample_labels(label, self.batch_size_per_image, self.positive_fraction, 0, PERM_FN)
        label.fill_(-1)
        label.scatter_(0, pos_idx, 1)
        label.scatter_(0, neg_idx, 0)
        return label

    def _decode_proposals(self, anchors: List[Boxes], pred_anchor_deltas: List[torch.Tensor]):
        N = pred_anchor_deltas[0].shape[0]
        proposals = []
        for anchors_i, deltas_i in zip(anchors, pred_anchor_deltas):
            B = anchors_i.tensor.size(1)
            deltas_i = deltas_i.reshape(-1, B)
            a = anchors_i.tensor.unsqueeze(0).expand(N, -1, -1).reshape(-1, B)
            proposals.append(self.box2box_transform.apply_deltas(deltas_i, a).view(N, -1, B))
        return proposals


def build_proposal_generator(cfg, input_shape):
    return PROPOSAL_GENERATOR_REGISTRY.get(cfg.MODEL.PROPOSAL_GENERATOR.NAME)(cfg, input_shape)


# ----------------------------------------------------------------------------- ROI heads
class ROIPooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224, canonical_level=4):
        super().__init__()
        assert pooler_type == "ROIAlignV2" and sampling_ratio == 0 and len(scales) == 1
        self.output_size = output_size if isinstance(output_size, int) else output_size[0]
        self.scale = scales[0]

    def forward(self, x: List[torch.Tensor], box_lists: List[Boxes]):
        rois = d2.convert_boxes_to_pooler_format(box_lists)
        return d2.roi_align(x[0], rois, self.output_size, self.scale)


class FastRCNNConvFCHead(nn.Sequential):
    @configurable
    def __init__(self, input_shape: ShapeSpec, *, conv_dims: List[int], fc_dims: List[int], conv_norm=""):
        super().__init__()
        assert len(conv_dims) == 0
        self._output_size = (input_shape.channels, input_shape.height, input_shape.width)
        self.fcs = []
        for k, fc_dim in enumerate(fc_dims):
            if k == 0:
                self.add_module("flatten", nn.Flatten())
            fc = nn.Linear(int(math.prod(self._output_size)) if not isinstance(self._output_size, int)
                           else self._output_size, fc_dim)
            self.add_module("fc{}".format(k + 1), fc)
            self.add_module("fc_relu{}".format(k + 1), nn.ReLU())
            self.fcs.append(fc)
            self._output_size = fc_dim
        for layer in self.fcs:
            d2.c2_xavier_fill(layer)

    @classmethod
    def from_config(cls, cfg, input_shape):
        H = cfg.MODEL.ROI_BOX_HEAD
        return {"input_shape": input_shape, "conv_dims": [H.CONV_DIM] * H.NUM_CONV, "fc_dims": [H.FC_DIM] * H.NUM_FC,
                "conv_norm": H.NORM}

    def forward(self, x):
        for layer in self:
            x = layer(x)
        return x

    @property
    def output_shape(self):
        o = self._output_size
        return ShapeSpec(channels=o) if isinstance(o, int) else ShapeSpec(channels=o[0], height=o[1], width=o[2])


ROI_BOX_HEAD_REGISTRY.register(FastRCNNConvFCHead)


def build_box_head(cfg, input_shape):
    return ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, input_shape)


class FastRCNNOutputLayers(nn.Module):
    @configurable
    def __init__(self, input_shape, *, box2box_transform, num_classes, test_score_thresh=0.0, test_nms_thresh=0.5,
                 test_topk_per_image=100, cls_agnostic_bbox_reg=False, smooth_l1_beta=0.0,
                 box_reg_loss_type="smooth_l1", loss_weight=1.0):
        super().__init__()
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        self.num_classes = num_classes
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.cls_score = nn.Linear(input_size, num_classes + 1)
        num_bbox_reg_classes = 1 if cls_agnostic_bbox_reg else num_classes
        box_dim = len(box2box_transform.weights)
        self.bbox_pred = nn.Linear(input_size, num_bbox_reg_classes * box_dim)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in [self.cls_score, self.bbox_pred]:
            nn.init.constant_(l.bias, 0)
        self.box2box_transform = box2box_transform
        self.smooth_l1_beta = smooth_l1_beta
        self.test_score_thresh = test_score_thresh
        self.test_nms_thresh = test_nms_thresh
        self.test_topk_per_image = test_topk_per_image
        self.box_reg_loss_type = box_reg_loss_type
        if isinstance(loss_weight, float):
            loss_weight = {"loss_cls": loss_weight, "loss_box_reg": loss_weight}
        self.loss_weight = loss_weight

    @classmethod
    def from_config(cls, cfg, input_shape):
        return {
            "input_shape": input_shape, "box2box_transform": None, "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "cls_agnostic_bbox_reg": cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG,
            "smooth_l1_beta": cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA,
            "test_score_thresh": cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
            "test_nms_thresh": cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST,
            "test_topk_per_image": cfg.TEST.DETECTIONS_PER_IMAGE,
            "box_reg_loss_type": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE,
            "loss_weight": {"loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT},
        }

    def forward(self, x):
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        return self.cls_score(x), self.bbox_pred(x)

    def losses(self, predictions, proposals):
        scores, proposal_deltas = predictions
        gt_classes = cat([p.gt_classes for p in proposals], dim=0) if len(proposals) else torch.empty(0)
        if len(proposals):
            proposal_boxes = cat([p.proposal_boxes.tensor for p in proposals], dim=0)
            gt_boxes = cat([(p.gt_boxes if p.has("gt_boxes") else p.proposal_boxes).tensor for p in proposals], dim=0)
        else:
            proposal_boxes = gt_boxes = torch.empty((0, 4), device=proposal_deltas.device)
        losses = {"loss_cls": d2.cross_entropy(scores, gt_classes, reduction="mean"),
                  "loss_box_reg": self.box_reg_loss(proposal_boxes, gt_boxes, proposal_deltas, gt_classes)}
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in losses.items()}


class ROIHeads(nn.Module):
    @configurable
    def __init__(self, *, num_classes, batch_size_per_image, positive_fraction, proposal_matcher, proposal_append_gt=True):
        super().__init__()
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.num_classes = num_classes
        self.proposal_matcher = proposal_matcher
        self.proposal_append_gt = proposal_append_gt

    @classmethod
    def from_config(cls, cfg):
        H = cfg.MODEL.ROI_HEADS
        return {"batch_size_per_image": H.BATCH_SIZE_PER_IMAGE, "positive_fraction": H.POSITIVE_FRACTION,
                "num_classes": H.NUM_CLASSES, "proposal_append_gt": H.PROPOSAL_APPEND_GT,
                "proposal_matcher": Matcher(H.IOU_THRESHOLDS, H.IOU_LABELS, allow_low_quality_matches=False)}

    def _sample_proposals(self, matched_idxs, matched_labels, gt_classes):
        has_gt = gt_classes.numel() > 0
        if has_gt:
            gt_classes = gt_classes[matched_idxs]
            gt_classes[matched_labels == 0] = self.num_classes
            gt_classes[matched_labels == -1] = -1
        else:
            gt_classes = torch.zeros_like(matched_idxs) + self.num_classes
        fg, bg = d2.subsample_labels(gt_classes, self.batch_size_per_image, self.positive_fraction,
                                     self.num_classes, PERM_FN)
        sampled = torch.cat([fg, bg], dim=0)
        return sampled, gt_classes[sampled]


class StandardROIHeads(ROIHeads):
    @configurable
    def __init__(self, *, box_in_features, box_pooler, box_head, box_predictor, train_on_pred_boxes=False, **kwargs):
        super().__init__(**kwargs)
        self.in_features = self.box_in_features = box_in_features
        self.box_pooler = box_pooler
        self.box_head = box_head
        self.box_predictor = box_predictor
        self.mask_on = self.keypoint_on = False
        self.train_on_pred_boxes = train_on_pred_boxes

    @classmethod
    def from_config(cls, cfg, input_shape):
        ret = super().from_config(cfg)
        ret["train_on_pred_boxes"] = cfg.MODEL.ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES
        if inspect.ismethod(cls._init_box_head):
            ret.update(cls._init_box_head(cfg, input_shape))
        return ret


def build_roi_heads(cfg, input_shape):
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape)


# ----------------------------------------------------------------------------- meta arch
def build_backbone(cfg, input_shape=None):
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)


class GeneralizedRCNN(nn.Module):
    @configurable
    def __init__(self, *, backbone, proposal_generator, roi_heads, pixel_mean, pixel_std, input_format=None, vis_period=0):
        super().__init__()
        self.backbone = backbone
        self.proposal_generator = proposal_generator
        self.roi_heads = roi_heads
        self.input_format = input_format
        self.vis_period = vis_period
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std).view(-1, 1, 1), False)

    @classmethod
    def from_config(cls, cfg):
        backbone = build_backbone(cfg)
        return {"backbone": backbone, "proposal_generator": build_proposal_generator(cfg, backbone.output_shape()),
                "roi_heads": build_roi_heads(cfg, backbone.output_shape()), "input_format": cfg.INPUT.FORMAT,
                "vis_period": cfg.VIS_PERIOD, "pixel_mean": cfg.MODEL.PIXEL_MEAN, "pixel_std": cfg.MODEL.PIXEL_STD}

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs):
        images = [x["image"].to(self.device) for x in batched_inputs]
        images = [(x - self.pixel_mean) / self.pixel_std for x in images]
        return ImageList.from_tensors(images, self.backbone.size_divisibility)


def build_model(cfg):
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
