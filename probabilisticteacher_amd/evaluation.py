"""Pascal-VOC style detection evaluation.

Reference: pt/engine/trainer.py:127-137 `build_evaluator` (TEST.EVALUATOR == "VOCeval" ->
detectron2.evaluation.PascalVOCDetectionEvaluator), driven from the eval hooks at trainer.py:529-542; the README's
mAP50 tables are this evaluator's "AP50".  detectron2 is not vendored under /root/reference: the protocol below restates
D2 0.5's pascal_voc_evaluation.py (itself the official VOCdevkit / py-faster-rcnn `voc_eval`): detections are written as
text with 3 (score) / 1 (coordinates) decimals and 1-based corners, matched greedily in descending score order against
the ground truth of their class with the `+ 1` pixel-inclusive IoU, "difficult" objects neither count nor punish, AP by
the VOC2010+ area rule or the VOC2007 11-point rule, for IoU thresholds 0.50:0.05:0.95.

Ground truth comes from the records themselves ("instances".gt_boxes / gt_classes [+ "difficult"]) instead of the
VOC XML files D2 parses: D2 subtracts 1 from xmin/ymin when it builds a training record, so the XML box is
[x1 + 1, y1 + 1, x2, y2] of the record's box."""
from collections import OrderedDict, defaultdict
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch


def voc_ap(rec: np.ndarray, prec: np.ndarray, use_07_metric: bool = False) -> float:
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.0
        return float(ap)
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def voc_eval(dets: List[tuple], gts: Dict[object, dict], ovthresh: float = 0.5, use_07_metric: bool = False):
    """dets: [(image_id, score, x1, y1, x2, y2)] of ONE class; gts: image_id -> {"bbox": (M,4), "difficult": (M,) bool}.
    Returns (rec, prec, ap)."""
    class_recs, npos = {}, 0
    for iid, g in gts.items():
        diff = np.asarray(g["difficult"], dtype=bool)
        class_recs[iid] = {"bbox": np.asarray(g["bbox"], dtype=float).reshape(-1, 4), "difficult": diff,
                           "det": [False] * len(diff)}
        npos += int(np.sum(~diff))
    image_ids = [d[0] for d in dets]
    confidence = np.array([float(d[1]) for d in dets])
    BB = np.array([[float(z) for z in d[2:]] for d in dets]).reshape(-1, 4)
    order = np.argsort(-confidence)
    BB = BB[order, :]
    image_ids = [image_ids[x] for x in order]
    nd = len(image_ids)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d in range(nd):
        R = class_recs.get(image_ids[d], {"bbox": np.zeros((0, 4)), "difficult": np.zeros(0, bool), "det": []})
        bb = BB[d, :].astype(float)
        ovmax = -np.inf
        BBGT = R["bbox"].astype(float)
        if BBGT.size > 0:
            ixmin = np.maximum(BBGT[:, 0], bb[0])
            iymin = np.maximum(BBGT[:, 1], bb[1])
            ixmax = np.minimum(BBGT[:, 2], bb[2])
            iymax = np.minimum(BBGT[:, 3], bb[3])
            iw = np.maximum(ixmax - ixmin + 1.0, 0.0)
            ih = np.maximum(iymax - iymin + 1.0, 0.0)
            inters = iw * ih
            uni = ((bb[2] - bb[0] + 1.0) * (bb[3] - bb[1] + 1.0)
                   + (BBGT[:, 2] - BBGT[:, 0] + 1.0) * (BBGT[:, 3] - BBGT[:, 1] + 1.0) - inters)
            overlaps = inters / uni
            ovmax = np.max(overlaps)
            jmax = int(np.argmax(overlaps))
        if ovmax > ovthresh:
            if not R["difficult"][jmax]:
                if not R["det"][jmax]:
                    tp[d] = 1.0
                    R["det"][jmax] = True
                else:
                    fp[d] = 1.0
        else:
            fp[d] = 1.0
    fp = np.cumsum(fp)
    tp = np.cumsum(tp)
    rec = tp / float(npos) if npos > 0 else tp * 0.0
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


class PascalVOCDetectionEvaluator:
    """process(inputs, outputs) per batch, evaluate() -> OrderedDict(bbox={"AP", "AP50", "AP75"}) (values in percent)."""

    def __init__(self, class_names: Sequence[str], is_2007: bool = False):
        self._class_names = list(class_names)
        self._is_2007 = bool(is_2007)
        self.reset()

    def reset(self):
        self._predictions = defaultdict(list)         # class id -> [(image_id, score, x1, y1, x2, y2)]
        self._gt = {}                                 # image_id -> (boxes (M,4) in XML convention, classes, difficult)

    def process(self, inputs: Iterable[dict], outputs: Iterable[dict]):
        for inp, out in zip(inputs, outputs):
            image_id = inp.get("image_id", len(self._gt))
            if "instances" in inp and inp["instances"].has("gt_boxes"):
                gt = inp["instances"]
                b = gt.gt_boxes.tensor.detach().cpu().numpy().astype(float).copy()
                b[:, 0] += 1.0
                b[:, 1] += 1.0
                diff = gt.difficult.cpu().numpy().astype(bool) if gt.has("difficult") else np.zeros(len(b), bool)
                self._gt[image_id] = (b, gt.gt_classes.cpu().numpy(), diff)
            else:
                self._gt.setdefault(image_id, (np.zeros((0, 4)), np.zeros(0, np.int64), np.zeros(0, bool)))
            inst = out["instances"]
            boxes = inst.pred_boxes.tensor.detach().cpu().numpy()
            scores = inst.scores.detach().cpu().tolist()
            classes = inst.pred_classes.detach().cpu().tolist()
            for box, score, cls in zip(boxes, scores, classes):
                xmin, ymin, xmax, ymax = box
                # the (matlab) VOC toolkit takes 1-based corners, written with fixed precision (D2 keeps the text round trip)
                line = f"{score:.3f} {xmin + 1:.1f} {ymin + 1:.1f} {xmax:.1f} {ymax:.1f}".split(" ")
                self._predictions[cls].append((image_id,) + tuple(float(v) for v in line))

    def _gather(self):
        """D2's evaluator gathers every rank's predictions on rank 0 before scoring (comm.gather of the pickled lists); the
        ground truth of the images a rank processed travels with them.  Non-zero ranks return None from evaluate()."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return True
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, (dict(self._predictions), self._gt))
        if dist.get_rank() != 0:
            return False
        self._predictions, self._gt = defaultdict(list), {}
        for preds, gt in parts:
            for c, lines in preds.items():
                self._predictions[c].extend(lines)
            self._gt.update(gt)
        return True

    def evaluate(self) -> "Optional[OrderedDict[str, dict]]":
        if not self._gather():
            return None
        aps = defaultdict(list)                       # iou threshold (percent) -> per-class APs
        for cls_id, _ in enumerate(self._class_names):
            dets = self._predictions.get(cls_id, [])
            gts = {iid: {"bbox": b[c == cls_id], "difficult": d[c == cls_id]} for iid, (b, c, d) in self._gt.items()}
            for thresh in range(50, 100, 5):
                _, _, ap = voc_eval(dets, gts, ovthresh=thresh / 100.0, use_07_metric=self._is_2007)
                aps[thresh].append(ap * 100)
        ret = OrderedDict()
        mAP = {iou: float(np.mean(x)) for iou, x in aps.items()}
        ret["bbox"] = {"AP": float(np.mean(list(mAP.values()))), "AP50": mAP[50], "AP75": mAP[75]}
        ret["per_class_AP50"] = {n: aps[50][i] for i, n in enumerate(self._class_names)}
        return ret


@torch.no_grad()
def inference_on_dataset(model, data_loader: Iterable[List[dict]], evaluator: PascalVOCDetectionEvaluator):
    """D2 inference_on_dataset: eval mode, `model(batched_inputs)` (rcnn.py:33-34 -> inference + detector_postprocess),
    evaluator.process per batch; the model's previous mode is restored."""
    was_training = model.training
    model.eval()
    evaluator.reset()
    try:
        for batch in data_loader:
            evaluator.process(batch, model(batch))
    finally:
        model.train(was_training)
    return evaluator.evaluate()
