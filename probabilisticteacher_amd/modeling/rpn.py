"""Gaussian RPN on HIP kernels (reference pt/modeling/proposal_generator/{rpn.py,proposal_utils.py}).

What runs where: the 3x3/1x1 head convs (MFMA), anchor grid, IoU + Matcher, box codec, segmented sorts,
proposal clipping/rescoring, NMS and every loss (+ its gradient) are HIP kernels from libptmi355.so; torch
only gathers/scatters rows by index (autograd glue) and carries per-image Python containers.

Reference quirks kept on purpose: 8-dim (mu, sigma-logit) deltas always; sigma rows of the
top-k proposals taken from the FIRST k anchors in raster order (proposal_utils.py:94); rescoring multiplies raw
logits (:136-138); RPN soft-label loss uses sigmoid(1-x) (rpn.py:299) and only positive anchors; the box target
mean_p is NOT detached in the RPN (rpn.py:315) so the learnable anchors receive gradient when danchor=True."""
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from .. import ops
from ..registry import PROPOSAL_GENERATOR_REGISTRY, RPN_HEAD_REGISTRY
from ..structures import Boxes, FreeInstances
from .anchor_generator import build_anchor_generator
from .box_regression import Box2BoxTransform
from . import sampling


class _ConvP(nn.Module):
    def __init__(self, cout, cin, k, std=0.01):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.normal_(self.weight, std=std)


@RPN_HEAD_REGISTRY.register()
class GuassianRPNHead(nn.Module):
    """StandardRPNHead with box_dim doubled (rpn.py:44-55): conv3x3+ReLU, 1x1 -> A logits, 1x1 -> A*8 deltas."""

    def __init__(self, cfg, input_shape):
        super().__init__()
        in_channels = input_shape[0].channels
        ag = build_anchor_generator(cfg, input_shape)
        num_anchors, box_dim = ag.num_anchors[0], ag.box_dim * 2
        self.conv = _ConvP(in_channels, in_channels, 3)
        self.objectness_logits = _ConvP(num_anchors, in_channels, 1)
        self.anchor_deltas = _ConvP(num_anchors * box_dim, in_channels, 1)

    def forward(self, features: List[torch.Tensor]):
        """D2's StandardRPNHead interface: lists of (N, A, H, W) logits and (N, 8A, H, W) deltas"""
        obj, deltas = [], []
        for x in features:
            t = ops.conv3x3(x, self.conv.weight, self.conv.bias, True)
            obj.append(ops.conv1x1(t, self.objectness_logits.weight, self.objectness_logits.bias))
            deltas.append(ops.conv1x1(t, self.anchor_deltas.weight, self.anchor_deltas.bias))
        return obj, deltas

    def forward_flat(self, features: List[torch.Tensor]):
        """What GuassianRPN.forward makes of `forward`'s result (rpn.py:97-113) -- logits (N, H W A) and deltas (N, H W A, 8)
        in anchor order -- computed in that layout directly (ops.rpn_head_1x1: the 1x1 convolutions as transposed GEMMs, no
        permute copies).  GuassianRPN uses this method when the head has it."""
        obj, deltas = [], []
        for x in features:
            t = ops.conv3x3(x, self.conv.weight, self.conv.bias, True)
            lg, d8 = ops.rpn_head_1x1(t, self.objectness_logits.weight, self.objectness_logits.bias,
                                      self.anchor_deltas.weight, self.anchor_deltas.bias)
            obj.append(lg)
            deltas.append(d8)
        return obj, deltas


def build_rpn_head(cfg, input_shape):
    return RPN_HEAD_REGISTRY.get(cfg.MODEL.RPN.HEAD_NAME)(cfg, input_shape)


def find_top_rpn_proposals(decoded, logits, sigma_logits, image_sizes, nms_thresh, pre_nms_topk, post_nms_topk,
                           min_box_size, training) -> List[FreeInstances]:
    """proposal_utils.py:27-154 for one feature level, batched over images on the device:
    segmented sort -> clip / nonempty / sigma-rescoring kernel (dropped entries get key -inf, kept counts stay on the
    device) -> segmented sort of the keys over fixed k-entry segments -> batched bitmask NMS on the first count[i]
    entries.  ONE device->host read per call (kept-after-NMS counts + the non-finite flags)."""
    n, r = logits.shape
    dev = logits.device
    k = min(r, pre_nms_topk)
    seg = torch.arange(0, (n + 1) * r, r, dtype=torch.int32, device=dev)
    srt, idx = ops.segsort_desc(logits.reshape(-1), seg, max_len=r, topk=k, lengths=(r,) * n)      # (rpn_prepare reads the first k of each row)
    sizes = torch.tensor([[float(h), float(w)] for h, w in image_sizes], dtype=torch.float32).pin_memory().to(
        dev, non_blocking=True)
    boxes, keys, counts, nonfinite = ops.rpn_prepare(decoded, srt.view(n, r), idx.view(n, r), sigma_logits, sizes, k,
                                                     float(min_box_size))
    seg2 = torch.arange(0, (n + 1) * k, k, dtype=torch.int32, device=dev)
    s2, i2 = ops.segsort_desc(keys.view(-1), seg2, max_len=k, lengths=(k,) * n)
    sb = torch.gather(boxes, 1, i2.view(n, k, 1).long().expand(n, k, 4)).view(n * k, 4)
    keep, kcnt = ops.nms_batched(sb, seg2, k, float(nms_thresh), int(post_nms_topk), seg_counts=counts)
    host = torch.cat([kcnt, nonfinite]).cpu().tolist()                # the one sync
    kc, bad = host[:n], host[n:]
    if any(bad) and training:
        raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
    # one gather for the whole batch (a per-image loop of cast + add + two index launches was 4 n tiny launches per call, issued
    # while the GPU had nothing else queued); the per-image results are row ranges of it
    keep_g = keep.long() + (torch.arange(n, device=dev) * k).unsqueeze(1)
    sel = torch.cat([keep_g[i, :kc[i]] for i in range(n)]) if n else keep_g.reshape(-1)
    boxes_all, logits_all = sb[sel], s2[sel]
    results, c0 = [], 0
    for i, size in enumerate(image_sizes):
        res = FreeInstances(size)
        res.proposal_boxes = Boxes(boxes_all[c0:c0 + kc[i]])
        res.objectness_logits = logits_all[c0:c0 + kc[i]]
        c0 += kc[i]
        results.append(res)
    return results


@PROPOSAL_GENERATOR_REGISTRY.register()
class GuassianRPN(nn.Module):
    def __init__(self, cfg, input_shape: Dict):
        super().__init__()
        R = cfg.MODEL.RPN
        self.cfg = cfg
        self.in_features = R.IN_FEATURES
        shapes = [input_shape[f] for f in self.in_features]
        # registration order as in D2's RPN.__init__ (head, then anchor generator): it fixes the state_dict key order
        # and, through D2's build_optimizer, the parameter numbering of the optimiser state in checkpoints
        self.rpn_head = build_rpn_head(cfg, shapes)
        self.anchor_generator = build_anchor_generator(cfg, shapes)
        self.box2box_transform = Box2BoxTransform(weights=R.BBOX_REG_WEIGHTS)
        self.iou_thresholds, self.iou_labels = list(R.IOU_THRESHOLDS), list(R.IOU_LABELS)
        self.batch_size_per_image = R.BATCH_SIZE_PER_IMAGE
        self.positive_fraction = R.POSITIVE_FRACTION
        self.pre_nms_topk = {True: R.PRE_NMS_TOPK_TRAIN, False: R.PRE_NMS_TOPK_TEST}
        self.post_nms_topk = {True: R.POST_NMS_TOPK_TRAIN, False: R.POST_NMS_TOPK_TEST}
        self.nms_thresh = R.NMS_THRESH
        self.min_box_size = float(cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE)
        self.anchor_boundary_thresh = R.BOUNDARY_THRESH
        self.loss_weight = {"loss_rpn_cls": R.LOSS_WEIGHT, "loss_rpn_loc": R.BBOX_REG_LOSS_WEIGHT * R.LOSS_WEIGHT}

    def head_outputs(self, feats):
        """the head's outputs in the flat anchor-order layout when the head offers it (no permute copies), else D2's"""
        flat = getattr(self.rpn_head, "forward_flat", None)
        return flat(feats) if flat is not None else self.rpn_head(feats)

    # ------------------------------------------------------------------ forward (rpn.py:80-154)
    @staticmethod
    def _flat_outputs(obj, deltas):
        if obj[0].dim() == 2:               # already (N, H*W*A) and (N, H*W*A, 8): GuassianRPNHead.forward_flat
            return obj[0], deltas[0]
        n, a, h, w = obj[0].shape
        # (N,A,H,W) -> (N,H*W*A);  (N,A*8,H,W) -> (N,H*W*A,8)   (rpn.py:97-113)
        return (obj[0].permute(0, 2, 3, 1).reshape(n, -1),
                deltas[0].view(n, a, 8, h, w).permute(0, 3, 4, 1, 2).reshape(n, -1, 8))

    def proposals_from_head(self, images, features, head_out):
        """the proposals of a whole batch from head outputs already computed (the joint student pass: ONE sort / NMS chain and ONE
        host read for its two branches; per image exactly what `forward` returns)"""
        feats = [features[f] for f in self.in_features]
        anchors = self.anchor_generator(feats)[0].tensor.detach()
        logits, d8 = self._flat_outputs(*head_out)
        return self.predict_proposals(anchors, logits, d8, images.image_sizes)

    def forward(self, images, features, gt_instances: Optional[List[FreeInstances]] = None, compute_loss=True,
                branch="", danchor=False, head_out=None, proposals=None):
        """`head_out` = (objectness, deltas) already computed by `self.rpn_head` on these features (the joint
        student pass runs the head once for both branches); `proposals` = this batch's proposals if they were already predicted
        (`proposals_from_head`)."""
        feats = [features[f] for f in self.in_features]
        assert len(feats) == 1, "single-level RPN (vgg_block5)"
        anchors = self.anchor_generator(feats)[0].tensor
        if not danchor:
            anchors = anchors.detach()      # grad_zero (rpn.py:91-94): the anchor table gets an all-zero gradient
        obj, deltas = head_out if head_out is not None else self.head_outputs(feats)
        logits, d8 = self._flat_outputs(obj, deltas)

        if branch == "unsupervised":
            losses = self._losses_unsup(anchors, logits, d8, gt_instances)
        elif self.training and compute_loss:
            losses = self._losses_sup(anchors, logits, d8, gt_instances)
            # the reference weights the supervised RPN losses TWICE -- inside `losses` (rpn.py:254) and again here
            # (rpn.py:141) -- and the unsupervised ones not at all (rpn.py:347-360); invisible at the shipped 1.0
            losses = {k: v * self.loss_weight.get(k, 1.0) * self.loss_weight.get(k, 1.0) for k, v in losses.items()}
        else:
            losses = {}
        if proposals is None:
            proposals = self.predict_proposals(anchors, logits, d8, images.image_sizes)
        return proposals, losses

    @torch.no_grad()
    def predict_proposals(self, anchors, logits, d8, image_sizes):
        n, r = logits.shape
        dd = d8.detach().contiguous()
        decoded = ops.apply_deltas(dd.view(n * r, 8), anchors.detach(), self.box2box_transform.weights,
                                   self.box2box_transform.scale_clamp, k=1, dstride=8).view(n, r, 4)
        sigma = dd[..., 4:].contiguous()
        return find_top_rpn_proposals(decoded, logits.detach().contiguous(), sigma, image_sizes, self.nms_thresh,
                                      self.pre_nms_topk[self.training], self.post_nms_topk[self.training],
                                      self.min_box_size, self.training)

    # ------------------------------------------------------------------ supervised (rpn.py:191-255, 363-448)
    def _losses_sup(self, anchors, logits, d8, gt_instances):
        n, r = logits.shape
        anc = anchors.detach()
        with torch.no_grad():
            # whole batch: one IoU-match launch pair, one sync-free relabel, one nonzero
            counts = [len(inst.gt_boxes) for inst in gt_instances]
            gt_all = torch.cat([inst.gt_boxes.tensor for inst in gt_instances], 0)
            midx, lab_all, _, gt_off, _ = ops.iou_match_batched(gt_all, counts, anc, None, self.iou_thresholds,
                                                                self.iou_labels, True)
            lab_all = sampling.keyed_relabel(lab_all, self.batch_size_per_image, self.positive_fraction, 0)
            flat_pos = torch.nonzero(lab_all.view(-1) == 1).squeeze(1)
            img = torch.div(flat_pos, r, rounding_mode="floor")
            # (positives only exist for images with ground truth, so the gather index is always in range)
            gt_rows = gt_all[midx.view(-1)[flat_pos] + gt_off[img].long()]
        inv = 1.0 / (self.batch_size_per_image * n)
        loss_cls = ops.bce_logits_sum(logits.contiguous(), lab_all, inv)
        d_rows = d8.reshape(-1, 8)[flat_pos]
        a_rows = anchors[flat_pos % r]
        tgt = self.box2box_transform.get_deltas(a_rows, gt_rows)
        loss_loc = ops.gaussian_nll_sum(d_rows, tgt, inv)
        return {"loss_rpn_cls": loss_cls, "loss_rpn_loc": loss_loc}

    # ------------------------------------------------------------------ unsupervised (rpn.py:257-361, 426-430)
    def _losses_unsup(self, anchors, logits, d8, pseudo: List[FreeInstances]):
        n, r = logits.shape
        U = self.cfg.UNSUPNET
        anc = anchors.detach()
        has_box = pseudo[0].has("boxes_sigma")
        with torch.no_grad():
            # one IoU match per image (different pseudo-box counts), then ONE nonzero for the whole batch: the
            # positives of image i select rows of the concatenated pseudo-label tensors through per-image offsets
            counts = [len(inst.pseudo_boxes) for inst in pseudo]
            pb_all = torch.cat([inst.pseudo_boxes.tensor for inst in pseudo], 0)
            midx, lab_all, _, gt_off, _ = ops.iou_match_batched(pb_all, counts, anc, None, self.iou_thresholds,
                                                                self.iou_labels, True)
            flat = torch.nonzero(lab_all.view(-1) == 1).squeeze(1)                 # positives only, no subsampling
            img = torch.div(flat, r, rounding_mode="floor")
            sel = midx.view(-1)[flat] + gt_off[img].long()
            all_logits = torch.cat([inst.scores_logists for inst in pseudo], 0)
            T = all_logits[sel].contiguous()
            sig_all = torch.cat([(inst.boxes_sigma if has_box else inst.scores_logists) for inst in pseudo], 0)[sel]
            tgt_all = pb_all[sel]
        inv = 1.0 / (self.batch_size_per_image * n)
        x = logits.reshape(-1)[flat]
        loss_cls, fg = ops.rpn_soft_obj_loss(T, x, U.TAU[0], U.EFL_LAMBDA[0], bool(U.EFL), inv)
        out = {"loss_rpn_cls": loss_cls}
        if has_box:
            q = d8.reshape(-1, 8)[flat]
            mu_p = self.box2box_transform.get_deltas(anchors[flat % r], tgt_all)
            out["loss_rpn_loc"] = ops.kl_efl_loss(q, mu_p, sig_all, fg, U.TAU[1], U.EFL_LAMBDA[1],
                                                  bool(U.EFL), 0, inv)
        return out


def build_proposal_generator(cfg, input_shape):
    return PROPOSAL_GENERATOR_REGISTRY.get(cfg.MODEL.PROPOSAL_GENERATOR.NAME)(cfg, input_shape)
