"""Gaussian ROI head on HIP kernels (reference pt/modeling/roi_heads/{roi_heads.py,fast_rcnn.py}).

ROIAlign, the FC box head (fp32 MFMA GEMMs with fused bias/ReLU), IoU matching, box codec, per-class NMS and all
losses (+gradients) are HIP kernels; torch gathers rows by index and carries the per-image containers.
Quirks kept: bbox_pred emits 8 numbers per class always; apply_deltas decodes the sigma
quadruples as boxes and they are dropped (fast_rcnn.py:63); teacher scores are multiplied by 1-mean(sigmoid(sigma))
(:101); per-class NMS uses torchvision's fp32 `boxes + cls*(max+1)` offset; unsup cls loss divides by R (NaN if 0)."""
import math
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from .. import ops
from ..registry import ROI_BOX_HEAD_REGISTRY, ROI_HEADS_REGISTRY
from ..structures import Boxes, FreeInstances
from .box_regression import Box2BoxTransform
from . import sampling

GT_LOGIT = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))     # proposal_utils.py:207


class _LinearP(nn.Module):
    def __init__(self, nin, nout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(nout, nin))
        self.bias = nn.Parameter(torch.zeros(nout))


class ROIPooler(nn.Module):
    """Single-level ROIAlignV2 pooler (D2 ROIPooler; constructed at roi_heads.py:68-73)."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type):
        super().__init__()
        assert pooler_type == "ROIAlignV2" and sampling_ratio == 0 and len(scales) == 1
        self.output_size, self.scale = int(output_size), float(scales[0])

    def forward(self, x: List[torch.Tensor], box_lists: List[Boxes]) -> torch.Tensor:
        dev = x[0].device
        counts = [len(b) for b in box_lists]
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c)
        if not box_lists or offs[-1] == 0:
            rois = torch.zeros((0, 5), device=dev)
        else:
            # the image-index column in one repeat_interleave (a `full` + `cat` pair per image was 2 n tiny launches); only the n
            # counts travel from the host (a fixed-size pinned staging block: no host allocation in the step)
            img_col = torch.repeat_interleave(torch.arange(len(counts), dtype=torch.float32, device=dev),
                                              ops.dev_i32(counts, dev).long(), output_size=offs[-1])
            rois = torch.cat([img_col.unsqueeze(1), torch.cat([b.tensor for b in box_lists], 0)], 1)
        # (ops.dev_i32: pinned + asynchronous -- torch.tensor(list, device=...) copies from pageable memory, which waits for the stream)
        img_offsets = ops.dev_i32(offs, dev) if len(box_lists) == x[0].shape[0] else None
        if (img_offsets is not None and offs[-1] > 0 and ops._native_bf16() and x[0].is_cuda
                and ops.roi_align_p8m_fits(x[0].shape[1], x[0].shape[2], x[0].shape[3], self.output_size)):
            # SOLVER.AMP.ENABLED: the pooling is deferred to its consumer -- the box head's first Linear layer takes its bf16
            # operands straight from the ROIAlign kernel (p8._RoiAlignLinearP8)
            return DeferredROIAlign(x[0], rois.contiguous(), img_offsets, self.output_size, self.scale)
        return ops.roi_align(x[0], rois.contiguous(), self.output_size, self.scale, img_offsets)


class DeferredROIAlign:
    """What ROIPooler hands to the box head in the bf16-storage mode: the ROIAlign call, not yet made."""

    def __init__(self, feat, rois, img_offsets, output_size, scale):
        self.feat, self.rois, self.img_offsets, self.output_size, self.scale = feat, rois, img_offsets, output_size, scale

    def materialize(self) -> torch.Tensor:
        return ops.roi_align(self.feat, self.rois, self.output_size, self.scale, self.img_offsets)


@ROI_BOX_HEAD_REGISTRY.register()
class FastRCNNConvFCHead(nn.Module):
    """flatten -> fc1 + ReLU -> fc2 + ReLU (D2 FastRCNNConvFCHead), c2_xavier_fill init."""

    def __init__(self, cfg, input_shape):
        super().__init__()
        H = cfg.MODEL.ROI_BOX_HEAD
        assert H.NUM_CONV == 0, "conv layers in the box head are not on the hot path"
        dim = input_shape.channels * input_shape.height * input_shape.width
        self.num_fc = H.NUM_FC
        for k in range(H.NUM_FC):
            fc = _LinearP(dim, H.FC_DIM)
            nn.init.kaiming_uniform_(fc.weight, a=1)
            setattr(self, f"fc{k + 1}", fc)
            dim = H.FC_DIM
        self.output_size = dim

    accepts_deferred_roi_align = True      # forward() takes ROIPooler's DeferredROIAlign (SOLVER.AMP.ENABLED) as well as a tensor

    def forward(self, x):
        first = 0
        if isinstance(x, DeferredROIAlign):
            from .. import p8
            if self.num_fc and self.fc1.weight.shape[1] >= p8.LINEAR_MIN_K:
                x = p8.roi_align_linear(x.feat, x.rois, x.img_offsets, x.output_size, x.scale, self.fc1.weight, self.fc1.bias, True)
                first = 1
            else:
                x = x.materialize()
        if first == 0:
            x = x.flatten(1)
        for k in range(first, self.num_fc):
            fc = getattr(self, f"fc{k + 1}")
            x = ops.linear(x, fc.weight, fc.bias, True)
        return x


def build_box_head(cfg, input_shape):
    return ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, input_shape)


class GuassianFastRCNNOutputLayers(nn.Module):
    """cls_score (K+1) and bbox_pred (8K) + Gaussian / entropy-focal losses + teacher inference
    (fast_rcnn.py:145-409)."""

    def __init__(self, cfg, input_size: int):
        super().__init__()
        self.cfg = cfg
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        assert not cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG
        self.model_type = cfg.UNSUPNET.MODEL_TYPE
        self.box2box_transform = Box2BoxTransform(weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS)
        self.cls_score = _LinearP(input_size, self.num_classes + 1)
        self.bbox_pred = _LinearP(input_size, self.num_classes * 8)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        self.test_score_thresh = cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST
        self.test_nms_thresh = cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST
        self.test_topk_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        self.loss_weight = {"loss_cls": 1.0, "loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT}

    def forward(self, x):
        return (ops.linear(x, self.cls_score.weight, self.cls_score.bias, False),
                ops.linear(x, self.bbox_pred.weight, self.bbox_pred.bias, False))

    # ---- supervised (D2 FastRCNNOutputLayers.losses + fast_rcnn.py:265-336)
    def losses(self, predictions, proposals: List[FreeInstances]):
        scores, deltas = predictions
        K = self.num_classes
        gt_classes = torch.cat([p.gt_classes for p in proposals], 0)
        pb = torch.cat([p.proposal_boxes.tensor for p in proposals], 0)
        gb = torch.cat([p.gt_boxes.tensor for p in proposals], 0)
        loss_cls = ops.softmax_ce_mean(scores, gt_classes)
        fg = torch.nonzero((gt_classes >= 0) & (gt_classes < K)).squeeze(1)
        d = deltas.view(-1, K, 8)[fg, gt_classes[fg]]
        tgt = self.box2box_transform.get_deltas(pb[fg], gb[fg])
        loss_box = ops.gaussian_nll_sum(d, tgt, 1.0 / max(gt_classes.numel(), 1.0))
        out = {"loss_cls": loss_cls, "loss_box_reg": loss_box}
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in out.items()}

    # ---- unsupervised (roi_heads.py:130-171 + fast_rcnn.py:179-263)
    def losses_unsupervised(self, predictions, proposals: List[FreeInstances]):
        scores, deltas = predictions
        K, U = self.num_classes, self.cfg.UNSUPNET
        T = torch.cat([p.soft_label for p in proposals]).detach().contiguous()
        r = T.shape[0]
        inv = 1.0 / r if r > 0 else float("nan")                      # fast_rcnn.py:209 divides by R
        out = {"loss_cls": ops.soft_ce_efl(T, scores, U.TAU[0], U.EFL_LAMBDA[0], bool(U.EFL), inv)}
        if proposals[0].has("boxes_sigma"):
            with torch.no_grad():
                sig_p = torch.cat([p.boxes_sigma for p in proposals])
                pb = torch.cat([p.proposal_boxes.tensor for p in proposals])
                psb = torch.cat([p.pseudo_boxes.tensor for p in proposals])
                cls = T.max(-1)[1] if r > 0 else torch.zeros(0, dtype=torch.int64, device=T.device)
                rows = torch.nonzero(cls != K).squeeze(1)
                mu_p = self.box2box_transform.get_deltas(pb[rows], psb[rows])
                sig_sel = sig_p[rows].contiguous()
            q = deltas.view(-1, K, 8)[rows, cls[rows]]                 # one gather instead of the per-row loop (:159-161)
            out["loss_box_reg"] = ops.kl_efl_loss(q, mu_p, sig_sel, None, U.TAU[1], U.EFL_LAMBDA[1], bool(U.EFL), 1, 1.0)
        return out

    # ---- teacher inference (fast_rcnn.py:338-409, :34-141)
    @torch.no_grad()
    def inference(self, predictions, proposals: List[FreeInstances]):
        """fast_rcnn.py:338-409 + fast_rcnn_inference(_single_image) :34-141 for the whole batch at once.
        `ptmi_roi_infer_prepare` does the per-ROI part (decode, finite filter, clip, score threshold, sigma rescoring,
        per-image candidate count / max coordinate) in one launch and leaves dense (roi, class) arrays; candidates are
        never compacted on the host: a segmented stable sort of the keys over the fixed (R_i * K)-entry segments puts
        each image's candidates first, in the order batched_nms visits them, and the NMS runs on the first count[i]
        entries (counts stay on the device).  ONE device->host read per call (the kept-detection counts)."""
        scores, deltas = predictions
        K = self.num_classes
        dev = scores.device
        n = len(proposals)
        counts = [len(p) for p in proposals]
        R = sum(counts)
        pb = torch.cat([p.proposal_boxes.tensor for p in proposals], 0).contiguous()
        probs = ops.softmax_rows(scores)
        roff = [0]
        for c in counts:
            roff.append(roff[-1] + c)
        roi_img = torch.repeat_interleave(torch.arange(n, dtype=torch.int32, device=dev), ops.dev_i32(counts, dev),
                                          output_size=R)
        hw = torch.tensor([[float(p.image_size[0]), float(p.image_size[1])] for p in proposals]).pin_memory().to(
            dev, non_blocking=True)
        boxes, keys, roi_valid, img_max, img_cnt, img_inv = ops.roi_infer_prepare(
            deltas.contiguous(), pb, probs, roi_img, hw, K, self.box2box_transform.weights,
            self.box2box_transform.scale_clamp, self.test_score_thresh)
        seg = ops.dev_i32([K * o for o in roff], dev)                     # dense (roi, class) segments per image
        # NMS capacity = the dense segment length K * max R_i (no host read to learn the candidate count).  Memory law of the
        # bitmask workspace: images * cap * ceil(cap / 64) * 8 bytes -- 32 MB per image at K = 8, R = 2000 (the shipped
        # configs, ~0.5 GB for 16 images), ~200 MB per image at K = 20, ~3.2 GB per image at K = 80; the scan kernel keeps one
        # row of the mask in LDS, which caps cap at 524 288 candidates per image.
        cap = K * max(counts) if counts else 0
        if cap > 524288:
            raise ValueError(f"ROI inference: {K} classes x {max(counts)} proposals = {cap} NMS candidates per image exceed the "
                             "524 288 the single-pass bitmask NMS holds; lower MODEL.RPN.POST_NMS_TOPK_TEST")
        srt, order = ops.segsort_desc(keys.view(-1), seg, max_len=cap, lengths=[K * c for c in counts])
        nms_boxes = ops.roi_infer_nms_boxes(boxes, order, seg, img_max, cap, K)
        topk = self.test_topk_per_image if self.test_topk_per_image >= 0 else max(cap, 1)
        keep, kcnt = ops.nms_batched(nms_boxes, seg, cap, float(self.test_nms_thresh), int(max(topk, 1)),
                                     seg_counts=img_cnt)
        host = torch.cat([kcnt, img_inv]).cpu().tolist()                  # the one sync
        kc, dropped = host[:n], host[n:]
        keep_g = keep.long() + seg[:-1].long().unsqueeze(1)                # (seg[i] = K * roff[i])
        pos = torch.cat([keep_g[i, :kc[i]] for i in range(n)], 0) if n else seg[:0].long()
        img_base = torch.repeat_interleave(seg[:-1].long(), ops.dev_i32(kc, dev).long(), output_size=sum(kc))
        dense = img_base + order[pos].long()                              # flat (roi, class) index of every detection
        roi, cls = torch.div(dense, K, rounding_mode="floor"), dense % K
        r_boxes, r_scores = boxes.view(-1, 4)[dense], srt[pos]
        r_sig = deltas.view(R, K, 8)[roi, cls, 4:]
        img_start = torch.div(img_base, K, rounding_mode="floor")        # first ROI row of the detection's image
        row = roi
        if any(dropped):
            # reference quirk (fast_rcnn.py:68-71,96,126): after the finite filter the kept ROIs are re-indexed by their
            # position in the FILTERED list, and `scores_logists` / the returned ROI indices use that position (the
            # logits are even read from the UNfiltered tensor at it)
            excl = torch.cumsum(roi_valid.long(), 0) - roi_valid.long()
            row = img_start + (excl[roi] - excl[img_start])
        r_logits = scores[row]
        results, kept_rows = [], []
        c0 = 0
        row_local = row - img_start                                        # ROI index within the image (img_start = roff[i])
        for i, prop in enumerate(proposals):
            k = kc[i]
            res = FreeInstances(prop.image_size)
            res.pred_boxes = Boxes(r_boxes[c0:c0 + k])
            res.scores = r_scores[c0:c0 + k]
            res.pred_classes = cls[c0:c0 + k]
            res.scores_logists = r_logits[c0:c0 + k]
            res.boxes_sigma = r_sig[c0:c0 + k]
            results.append(res)
            kept_rows.append(row_local[c0:c0 + k])
            c0 += k
        return results, kept_rows


@ROI_HEADS_REGISTRY.register()
class GuassianROIHead(nn.Module):
    def __init__(self, cfg, input_shape: Dict):
        super().__init__()
        self.cfg = cfg
        H = cfg.MODEL.ROI_HEADS
        self.num_classes = H.NUM_CLASSES
        self.batch_size_per_image = H.BATCH_SIZE_PER_IMAGE
        self.positive_fraction = H.POSITIVE_FRACTION
        self.proposal_append_gt = H.PROPOSAL_APPEND_GT
        self.iou_thresholds, self.iou_labels = list(H.IOU_THRESHOLDS), list(H.IOU_LABELS)
        self.box_in_features = H.IN_FEATURES
        B = cfg.MODEL.ROI_BOX_HEAD
        in_channels = input_shape[self.box_in_features[0]].channels
        scales = tuple(1.0 / input_shape[k].stride for k in self.box_in_features)
        self.box_pooler = ROIPooler(B.POOLER_RESOLUTION, scales, B.POOLER_SAMPLING_RATIO, B.POOLER_TYPE)
        from .backbone import ShapeSpec
        self.box_head = build_box_head(cfg, ShapeSpec(channels=in_channels, height=B.POOLER_RESOLUTION,
                                                      width=B.POOLER_RESOLUTION))
        self.box_predictor = GuassianFastRCNNOutputLayers(cfg, self.box_head.output_size)
        self.train_on_pred_boxes = B.TRAIN_ON_PRED_BOXES

    def forward(self, images, features, proposals, targets=None, compute_loss=True, branch=""):
        if self.training and compute_loss:
            assert targets
            proposals = self.label_and_sample_proposals(proposals, targets, branch=branch)
        feats = [features[f] for f in self.box_in_features]
        box_features = self.box_pooler(feats, [x.proposal_boxes for x in proposals])
        if isinstance(box_features, DeferredROIAlign) and not getattr(self.box_head, "accepts_deferred_roi_align", False):
            box_features = box_features.materialize()       # any other head registered in ROI_BOX_HEAD_REGISTRY gets a tensor
        predictions = self.box_predictor(self.box_head(box_features))
        del box_features
        if branch == "unsupervised" and self.training:
            return proposals, self.box_predictor.losses_unsupervised(predictions, proposals)
        if self.training and compute_loss:
            return proposals, self.box_predictor.losses(predictions, proposals)
        pred_instances, _ = self.box_predictor.inference(predictions, proposals)
        return pred_instances, predictions

    @torch.no_grad()
    def label_and_sample_proposals(self, proposals, targets, branch=""):
        """roi_heads.py:192-255 (+ proposal_utils.py:157-224 GT append, D2 _sample_proposals A.12) and
        roi_heads.py:257-291 for the unsupervised branch."""
        K = self.num_classes
        out = []
        dev = proposals[0].proposal_boxes.tensor.device
        n = len(proposals)
        if branch == "unsupervised":
            # matched-label-1 proposals only (roi_heads.py:257-291): one IoU-match launch pair, one count read and one
            # nonzero for the whole batch instead of a kernel chain + host sync per image
            pcounts = [len(p.proposal_boxes) for p in proposals]
            gcounts = [len(t.pseudo_boxes) for t in targets]
            pb_all = torch.cat([t.pseudo_boxes.tensor for t in targets], 0)
            boxes_all = torch.cat([p.proposal_boxes.tensor for p in proposals], 0)
            midx, mlab, _, gt_off, box_off = ops.iou_match_batched(pb_all, gcounts, boxes_all, pcounts, self.iou_thresholds,
                                                                   self.iou_labels, False)
            hit = mlab == 1
            img_of = torch.repeat_interleave(torch.arange(n, device=dev), box_off[1:] - box_off[:-1],
                                             output_size=boxes_all.shape[0])
            counts = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, img_of, hit.long()).cpu().tolist()
            flat = torch.nonzero(hit).squeeze(1)
            glob = midx[flat] + gt_off[img_of[flat]].long()                 # rows of the concatenated pseudo labels
            sel_boxes = boxes_all[flat]
            has_sigma = targets[0].has("boxes_sigma")
            if pb_all.shape[0] > 0:
                sel_pb = pb_all[glob]
                sel_soft = torch.cat([t.scores_logists for t in targets], 0)[glob]
                sel_sig = torch.cat([t.boxes_sigma for t in targets], 0)[glob] if has_sigma else None
            c0 = 0
            for prop, tgt, cnt, gc in zip(proposals, targets, counts, gcounts):
                r = FreeInstances(prop.image_size)
                r.proposal_boxes = Boxes(sel_boxes[c0:c0 + cnt])
                if gc == 0:
                    r.pseudo_boxes, r.soft_label = tgt.pseudo_boxes, tgt.scores_logists
                    if has_sigma:
                        r.boxes_sigma = tgt.boxes_sigma
                else:
                    r.pseudo_boxes, r.soft_label = Boxes(sel_pb[c0:c0 + cnt]), sel_soft[c0:c0 + cnt]
                    if has_sigma:
                        r.boxes_sigma = sel_sig[c0:c0 + cnt]
                c0 += cnt
                out.append(r)
            return out
        # the whole batch at once.  Proposals (+ appended ground truth) of all images are
        # concatenated; labels come from one batched IoU match, the 512-per-image sample from one sampling launch
        # (random keys, see sampling.py); ONE device->host read (the sample sizes) for the batch.
        gcounts = [len(t.gt_boxes) for t in targets]
        gt_all = torch.cat([t.gt_boxes.tensor for t in targets], 0)
        total_gt = gt_all.shape[0]
        parts_b, parts_l, bcounts = [], [], []
        gt_logit = torch.full((total_gt,), GT_LOGIT, device=dev)
        g0 = 0
        for prop, tgt, gc in zip(proposals, targets, gcounts):
            parts_b.append(prop.proposal_boxes.tensor)
            parts_l.append(prop.objectness_logits)
            cnt = len(prop.proposal_boxes)
            if self.proposal_append_gt:
                parts_b.append(tgt.gt_boxes.tensor)
                parts_l.append(gt_logit[g0:g0 + gc])
                cnt += gc
            g0 += gc
            bcounts.append(cnt)
        boxes_all = torch.cat(parts_b, 0)
        logits_all = torch.cat(parts_l, 0)
        midx, mlab, _, gt_off, box_off = ops.iou_match_batched(gt_all, gcounts, boxes_all, bcounts, self.iou_thresholds,
                                                               self.iou_labels, False)
        total = boxes_all.shape[0]
        img_of = torch.repeat_interleave(torch.arange(n, device=dev), box_off[1:] - box_off[:-1], output_size=total)
        if total_gt > 0:
            gidx = (midx + gt_off[img_of].long()).clamp_(max=total_gt - 1)          # images without gt: masked below
            cls = torch.cat([t.gt_classes for t in targets], 0)[gidx]
            cls = torch.where(mlab == 0, torch.full_like(cls, K), cls)              # also covers images without gt
            cls = torch.where(mlab == -1, torch.full_like(cls, -1), cls)
        else:
            gidx = None
            cls = torch.full((total,), K, dtype=torch.int64, device=dev)
        keys = sampling.draw_keys(cls, bcounts, K)
        npos = int(self.batch_size_per_image * self.positive_fraction)
        fg, bg, cnt = ops.sample_by_keys(cls, keys, box_off, max(bcounts), self.batch_size_per_image, npos, K)
        cnt_h = cnt.cpu().tolist()                                                   # the one sync
        fg = fg + box_off[:-1].long().unsqueeze(1)
        bg = bg + box_off[:-1].long().unsqueeze(1)
        sel = torch.cat([t for i, (nf, nb) in enumerate(cnt_h) for t in (fg[i, :nf], bg[i, :nb])], 0)
        sel_boxes, sel_logits, sel_cls = boxes_all[sel], logits_all[sel], cls[sel]
        if total_gt > 0:
            has_gt = torch.tensor([float(gc > 0) for gc in gcounts]).pin_memory().to(dev, non_blocking=True)
            sel_gt = gt_all[gidx[sel]] * has_gt[img_of[sel]].unsqueeze(1)            # zeros for images without gt
        else:
            sel_gt = boxes_all.new_zeros((sel.shape[0], 4))
        c0 = 0
        for prop, (nf, nb) in zip(proposals, cnt_h):
            k = nf + nb
            r = FreeInstances(prop.image_size)
            r.proposal_boxes = Boxes(sel_boxes[c0:c0 + k])
            r.objectness_logits = sel_logits[c0:c0 + k]
            r.gt_classes = sel_cls[c0:c0 + k]
            r.gt_boxes = Boxes(sel_gt[c0:c0 + k])
            c0 += k
            out.append(r)
        return out


def build_roi_heads(cfg, input_shape):
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape)
