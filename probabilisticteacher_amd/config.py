"""Configuration surface of the hot path: a small yacs-style CfgNode (yacs itself is not installed), the
detectron2==0.5 defaults the step reads and the reference's extra keys
(reference pt/config.py:20-92).  YAML files under configs/ use the same keys, `_BASE_` inheritance and
`KEY VALUE` command-line overrides as the reference's train_net.py / train.sh."""
from __future__ import annotations

import ast
import copy
import os
from typing import Any, List

import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"Attempted to set {k} to {v}, but CfgNode is immutable")
        self[k] = v

    def freeze(self):
        object.__setattr__(self, "_frozen", True)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        object.__setattr__(self, "_frozen", False)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def is_frozen(self):
        return object.__getattribute__(self, "_frozen")

    def clone(self):
        c = CfgNode(copy.deepcopy(dict(self)))
        return c

    def __deepcopy__(self, memo):
        return CfgNode({k: copy.deepcopy(v, memo) for k, v in self.items()})

    # ------------------------------------------------------------------ merging
    def _merge(self, other: dict, path: str = ""):
        for k, v in other.items():
            full = f"{path}.{k}" if path else k
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            if isinstance(v, dict):
                if not isinstance(self[k], CfgNode):
                    raise KeyError(f"{full} is not a config node")
                self[k]._merge(v, full)
            else:
                self[k] = _coerce(v, self[k], full)

    def merge_from_other_cfg(self, other: "CfgNode"):
        self._merge(other)

    def merge_from_file(self, path: str):
        self._merge(_load_yaml_with_base(path))

    def merge_from_list(self, opts: List[str]):
        assert len(opts) % 2 == 0, f"Override list has odd length: {opts}"
        for full, raw in zip(opts[0::2], opts[1::2]):
            node = self
            keys = full.split(".")
            for k in keys[:-1]:
                if k not in node:
                    raise KeyError(f"Non-existent config key: {full}")
                node = node[k]
            if keys[-1] not in node:
                raise KeyError(f"Non-existent config key: {full}")
            node[keys[-1]] = _coerce(_decode(raw), node[keys[-1]], full)

    def dump(self) -> str:
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v)
                    for k, v in n.items()}
        return yaml.safe_dump(plain(self), default_flow_style=None)


def _decode(v):
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, old, key):
    """yacs' type rule: the replacement must have the type of the default (tuple<->list, int->float allowed)."""
    if old is None or type(new) == type(old):
        return new
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        return list(new)
    if isinstance(old, float) and isinstance(new, int):
        return float(new)
    if isinstance(old, str) and new is None:
        return new
    raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) for config key: {key}")


def _load_yaml_with_base(path: str) -> dict:
    with open(path) as f:
        cfg = yaml.safe_load(f) or {}

    def tuples(d):
        for k, v in list(d.items()):
            if isinstance(v, dict):
                tuples(v)
            elif isinstance(v, str) and v.startswith("(") and v.endswith(")"):
                d[k] = ast.literal_eval(v)
    tuples(cfg)
    if BASE_KEY in cfg:
        base = cfg.pop(BASE_KEY)
        if not os.path.isabs(base):
            base = os.path.join(os.path.dirname(path), base)
        merged = _load_yaml_with_base(base)

        def deep(a, b):
            for k, v in b.items():
                if isinstance(v, dict) and isinstance(a.get(k), dict):
                    deep(a[k], v)
                else:
                    a[k] = v
        deep(merged, cfg)
        return merged
    return cfg


def get_cfg() -> CfgNode:
    """The detectron2==0.5 default values of every key the train step reads."""
    C = CfgNode()
    C.VERSION = 2
    C.OUTPUT_DIR = "./output"
    C.SEED = -1
    C.VIS_PERIOD = 0
    C.INPUT = CfgNode(dict(FORMAT="BGR", MIN_SIZE_TRAIN=(800,), MAX_SIZE_TRAIN=1333, MIN_SIZE_TEST=800,
                           MAX_SIZE_TEST=1333, RANDOM_FLIP="horizontal"))
    C.DATASETS = CfgNode(dict(TRAIN=(), TEST=()))
    C.DATALOADER = CfgNode(dict(NUM_WORKERS=4, FILTER_EMPTY_ANNOTATIONS=True))      # D2 default: True
    C.TEST = CfgNode(dict(DETECTIONS_PER_IMAGE=100, EVAL_PERIOD=0))
    C.SOLVER = CfgNode(dict(
        LR_SCHEDULER_NAME="WarmupMultiStepLR", MAX_ITER=40000, BASE_LR=0.001, MOMENTUM=0.9, NESTEROV=False,
        WEIGHT_DECAY=0.0001, WEIGHT_DECAY_NORM=0.0, GAMMA=0.1, STEPS=(30000,), WARMUP_FACTOR=1.0 / 1000,
        WARMUP_ITERS=1000, WARMUP_METHOD="linear", CHECKPOINT_PERIOD=5000, IMS_PER_BATCH=16, BIAS_LR_FACTOR=1.0,
        WEIGHT_DECAY_BIAS=0.0001, AMP=dict(ENABLED=False)))
    C.MODEL = CfgNode(dict(
        DEVICE="cuda", META_ARCHITECTURE="GeneralizedRCNN", MASK_ON=False, KEYPOINT_ON=False, LOAD_PROPOSALS=False,
        WEIGHTS="", PIXEL_MEAN=[103.530, 116.280, 123.675], PIXEL_STD=[1.0, 1.0, 1.0],
        BACKBONE=dict(NAME="build_resnet_backbone", FREEZE_AT=2),
        ANCHOR_GENERATOR=dict(NAME="DefaultAnchorGenerator", SIZES=[[32, 64, 128, 256, 512]],
                              ASPECT_RATIOS=[[0.5, 1.0, 2.0]], ANGLES=[[-90, 0, 90]], OFFSET=0.0),
        PROPOSAL_GENERATOR=dict(NAME="RPN", MIN_SIZE=0),
        RPN=dict(HEAD_NAME="StandardRPNHead", IN_FEATURES=["res4"], BOUNDARY_THRESH=-1, IOU_THRESHOLDS=[0.3, 0.7],
                 IOU_LABELS=[0, -1, 1], BATCH_SIZE_PER_IMAGE=256, POSITIVE_FRACTION=0.5, BBOX_REG_LOSS_TYPE="smooth_l1",
                 BBOX_REG_LOSS_WEIGHT=1.0, BBOX_REG_WEIGHTS=(1.0, 1.0, 1.0, 1.0), SMOOTH_L1_BETA=0.0, LOSS_WEIGHT=1.0,
                 PRE_NMS_TOPK_TRAIN=12000, PRE_NMS_TOPK_TEST=6000, POST_NMS_TOPK_TRAIN=2000, POST_NMS_TOPK_TEST=1000,
                 NMS_THRESH=0.7),
        ROI_HEADS=dict(NAME="Res5ROIHeads", NUM_CLASSES=80, IN_FEATURES=["res4"], IOU_THRESHOLDS=[0.5],
                       IOU_LABELS=[0, 1], BATCH_SIZE_PER_IMAGE=512, POSITIVE_FRACTION=0.25, SCORE_THRESH_TEST=0.05,
                       NMS_THRESH_TEST=0.5, PROPOSAL_APPEND_GT=True),
        ROI_BOX_HEAD=dict(NAME="", BBOX_REG_LOSS_TYPE="smooth_l1", BBOX_REG_LOSS_WEIGHT=1.0,
                          BBOX_REG_WEIGHTS=(10.0, 10.0, 5.0, 5.0), SMOOTH_L1_BETA=0.0, POOLER_RESOLUTION=14,
                          POOLER_SAMPLING_RATIO=0, POOLER_TYPE="ROIAlignV2", NUM_FC=0, FC_DIM=1024, NUM_CONV=0,
                          CONV_DIM=256, NORM="", CLS_AGNOSTIC_BBOX_REG=False, TRAIN_ON_PRED_BOXES=False),
    ))
    return C


def add_config(cfg: CfgNode) -> None:
    """The reference's extra keys (pt/config.py:20-92), same names and defaults."""
    _C = cfg
    _C.SOLVER.IMG_PER_BATCH_LABEL = 16
    _C.SOLVER.IMG_PER_BATCH_UNLABEL = 16
    _C.SOLVER.FACTOR_LIST = (1,)
    _C.SOLVER.REFERENCE_WORLD_SIZE = 1
    _C.SOLVER.REFERENCE_BATCH_SIZE = 0
    _C.DATASETS.TRAIN_LABEL = ("coco_2017_train",)
    _C.DATASETS.TRAIN_UNLABEL = ("coco_2017_train",)
    _C.DATASETS.CROSS_DATASET = True
    _C.TEST.EVALUATOR = "COCOeval"
    _C.UNSUPNET = CfgNode(dict(
        Trainer="pt", PSEUDO_BBOX_SAMPLE="all", TEACHER_UPDATE_ITER=1, BURN_UP_STEP=4000, EMA_KEEP_RATE=0.0,
        LOSS_WEIGHT_TYPE="standard", SOURCE_LOSS_WEIGHT=1.0, TARGET_UNSUP_LOSS_WEIGHT=1.0, GUASSIAN=True,
        TAU=[0.5, 0.5], EFL=True, EFL_LAMBDA=[0.5, 0.5], MODEL_TYPE="GUASSIAN"))
    _C.MODEL.VGG = CfgNode(dict(DEPTH=16, OUT_FEATURES=["vgg_block5"], NORM="None", CONV5_OUT_CHANNELS=512,
                                PRETRAIN="./vgg16_caffe.pth"))
    _C.MODEL.ANCHOR_GENERATOR.ANCHOR = [[[181.0193, 90.5097], [128.0000, 128.0000], [90.5097, 181.0193],
                                         [362.0387, 181.0193], [256.0000, 256.0000], [181.0193, 362.0387],
                                         [724.0773, 362.0387], [512.0000, 512.0000], [362.0387, 724.0773]]]
    _C.EMAMODEL = CfgNode(dict(SUP_CONSIST=True))


def setup_cfg(config_file: str = "", opts: List[str] = ()) -> CfgNode:
    """train_net.py:38-49 `setup`: defaults + add_config + yaml (+_BASE_) + command-line overrides, frozen."""
    cfg = get_cfg()
    add_config(cfg)
    if config_file:
        cfg.merge_from_file(config_file)
    cfg.merge_from_list(list(opts))
    cfg.freeze()
    return cfg
