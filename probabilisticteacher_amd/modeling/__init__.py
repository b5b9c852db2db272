"""Registered model components (importing this package registers them, like the reference's train_net.py:26-31)."""
from .anchor_generator import DefaultAnchorGenerator, DifferentiableAnchorGenerator  # noqa: F401
from .backbone import build_vgg_backbone  # noqa: F401
from .meta_arch import EnsembleTSModel, GuassianGeneralizedRCNN, build_model  # noqa: F401
from .roi_heads import FastRCNNConvFCHead, GuassianROIHead  # noqa: F401
from .rpn import GuassianRPN, GuassianRPNHead  # noqa: F401
