"""VGG16 conv backbone on the MFMA conv kernels (reference pt/modeling/backbone/vgg.py).

Parameter names match the reference (`vgg_block{b}.0.conv{k}.{weight,bias}`, vgg.py:44,55,89-92) so its
checkpoints / the vgg16_caffe.pth key map (vgg.py:130-145) apply unchanged.  Blocks 1..FREEZE_AT are frozen
(vgg.py:175-180): they run without recording autograd state, so their activations are never kept."""
import os
from collections import namedtuple

import torch
from torch import nn

from .. import ops
from ..registry import BACKBONE_REGISTRY

ShapeSpec = namedtuple("ShapeSpec", ["channels", "height", "width", "stride"], defaults=[None, None, None, None])

VGG_CFGS = {
    11: [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
    13: [64, 64, "M", 128, 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
    16: [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"],
    19: [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"],
}
_TORCHVISION_FEATURE_IDX = {16: [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]}


class _ConvParams(nn.Module):
    """Holds `weight` / `bias` of one 3x3 conv (c2_msra_fill init, vgg.py:61-63)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
        self.out_channels = cout


class VGGBlock(nn.Module):
    """[conv3x3 + bias + ReLU] x k, then MaxPool(2,2) unless it is the last block (vgg.py:36-72)."""

    def __init__(self, in_channels, channel_cfg, pool=True):
        super().__init__()
        self.pool = pool
        self.num_convs = len(channel_cfg)
        for i, cout in enumerate(channel_cfg):
            setattr(self, f"conv{i + 1}", _ConvParams(in_channels, cout))
            in_channels = cout
        self.out_channels = in_channels
        self.stride = 2

    def forward(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            params = []
            for i in range(self.num_convs):
                c = getattr(self, f"conv{i + 1}")
                params += [c.weight, c.bias]
            return ops.vgg_block(x, self.pool, params)            # one autograd node, fused backward chain
        fuse_pool = self.pool and not torch.is_grad_enabled()     # frozen block: full-res activation not needed
        for i in range(self.num_convs):
            c = getattr(self, f"conv{i + 1}")
            if fuse_pool and i == self.num_convs - 1:
                return ops.conv3x3_relu_pool_nograd(x, c.weight, c.bias)
            x = ops.conv3x3(x, c.weight, c.bias, True)
        if self.pool:
            x = ops.maxpool2x2(x)
        return x

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        return self


class VGG(nn.Module):
    def __init__(self, stages, out_features=None, pretrain=""):
        super().__init__()
        self._names = []
        stride = 1
        self._out_feature_strides, self._out_feature_channels = {}, {}
        for i, block in enumerate(stages):
            name = f"vgg_block{i + 1}"
            self.add_module(name, nn.Sequential(block))
            self._names.append(name)
            if name == "vgg_block5":
                self._out_feature_strides[name] = self._out_feature_strides["vgg_block4"]
            else:
                stride *= block.stride
                self._out_feature_strides[name] = stride
            self._out_feature_channels[name] = block.out_channels
        self._out_features = out_features or [self._names[-1]]
        if pretrain:
            self._load_pretrained(pretrain)

    def _load_pretrained(self, path):
        """vgg.py:127-152: torchvision `features.N` keys -> `vgg_block{b}.0.conv{k}`."""
        if not os.path.exists(path):
            raise FileNotFoundError(f"MODEL.VGG.PRETRAIN={path!r} not found (set it to '' for random init)")
        sd = torch.load(path, map_location="cpu")
        names = [f"{n}.0.conv{k + 1}" for n in self._names for k in range(getattr(self, n)[0].num_convs)]
        own = self.state_dict()
        for idx, name in zip(_TORCHVISION_FEATURE_IDX[16], names):
            for suffix in ("weight", "bias"):
                if f"{name}.{suffix}" in own:
                    own[f"{name}.{suffix}"].copy_(sd[f"features.{idx}.{suffix}"])

    @property
    def size_divisibility(self):
        return 0

    def _forward_p8(self, x):
        """SOLVER.AMP.ENABLED: the whole stack on bf16-storage (P8) tensors -- the image is converted once (3 channels padded to
        one 16-channel chunk), every layer reads and writes bf16 (probabilisticteacher_amd/p8.py), and each requested feature map
        is converted back to fp32 NCHW once (its gradient re-enters the stack through the same conversion)."""
        from .. import p8
        n, c, h, w = x.shape
        t = p8.from_nchw(ops._chk(x.contiguous(), name="image batch"))
        outputs = {}
        for name in self._names:
            blk = getattr(self, name)[0]
            params = []
            for i in range(blk.num_convs):
                cv = getattr(blk, f"conv{i + 1}")
                params += [cv.weight, cv.bias]
            trainable = torch.is_grad_enabled() and any(p.requires_grad for p in params)
            if trainable or t.requires_grad:
                t = p8.block(t, n, c, h, w, blk.pool, params)
            else:
                with torch.no_grad():
                    for i in range(blk.num_convs):
                        wt, b = params[2 * i], params[2 * i + 1]
                        # a block without a backward pass: its last convolution pools in its epilogue (the full-resolution
                        # activation is never written)
                        epi = 4 if (blk.pool and i == blk.num_convs - 1) else 1
                        t = p8.conv3x3_raw(t, p8.pack_weights(wt, 0), ops._chk(b.contiguous()), None, n, c, wt.shape[0], h, w, epi)
                        c = wt.shape[0]
            c = blk.out_channels
            if blk.pool:
                h, w = h // 2, w // 2
            if name in self._out_features:
                outputs[name] = p8._ToNCHW.apply(t, n, c, h, w)
        return outputs

    def forward(self, x):
        if ops._native_bf16():
            return self._forward_p8(x)
        outputs = {}
        for name in self._names:
            block = getattr(self, name)[0]
            frozen = not any(p.requires_grad for p in block.parameters())
            if frozen and not x.requires_grad:
                with torch.no_grad():
                    x = block(x)
            else:
                x = block(x)
            if name in self._out_features:
                outputs[name] = x
        return outputs

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}

    def freeze(self, freeze_at=0):
        for idx, name in enumerate(self._names, start=1):
            if freeze_at >= idx:
                getattr(self, name)[0].freeze()
        return self


@BACKBONE_REGISTRY.register()
def build_vgg_backbone(cfg, input_shape):
    depth = cfg.MODEL.VGG.DEPTH
    out_features = cfg.MODEL.VGG.OUT_FEATURES
    layout = VGG_CFGS[depth]
    max_stage = max({"vgg_block1": 1, "vgg_block2": 2, "vgg_block3": 3, "vgg_block4": 4, "vgg_block5": 5}[f]
                    for f in out_features)
    pools = [i for i, v in enumerate(layout) if v == "M"]
    stages, start, cin = [], 0, input_shape.channels
    for s in range(max_stage):
        chans = layout[start:pools[s]]
        stages.append(VGGBlock(cin, chans, pool=(s + 1 != 5)))
        cin, start = chans[-1], pools[s] + 1
    pretrain = cfg.MODEL.VGG.PRETRAIN if cfg.MODEL.VGG.PRETRAIN else ""
    return VGG(stages, out_features=out_features, pretrain=pretrain).freeze(cfg.MODEL.BACKBONE.FREEZE_AT)


def build_backbone(cfg, input_shape=None):
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)
