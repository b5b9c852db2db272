"""oracle/voc.py -- TEST INFRASTRUCTURE: an independent restatement of the Pascal-VOC detection protocol the reference's
evaluator applies (pt/engine/trainer.py:127-137 -> detectron2.evaluation.PascalVOCDetectionEvaluator, detectron2 0.5; not
vendored under /root/reference, restated from the VOCdevkit definition: parity unpinned by necessity).

Written separately from probabilisticteacher_amd/evaluation.py (plain Python lists and loops, no shared helpers) so that the
product evaluator has something other than itself to be checked against.  The protocol, in words:
  * a detection of class c in image i is a line `score x1 y1 x2 y2` with the score printed with 3 decimals and the corners as
    1-based pixels with 1 decimal (x1 + 1, y1 + 1, x2, y2 of the 0-based box) -- the numbers are read back from that text;
  * ground truth boxes are the XML's 1-based inclusive corners; "difficult" objects are neither positives nor punishments;
  * detections of a class are visited in descending score order (numpy's argsort of the negated scores IS the protocol's
    tie order), each is matched to the ground-truth box of its image with the largest pixel-inclusive IoU
    (intersection and areas with + 1); above the threshold the first detection of a box is a true positive, later ones are
    false positives, below it the detection is a false positive;
  * AP = area under the monotone precision envelope (VOC2010+) or the 11-point mean (VOC2007); the evaluator reports the
    mean over classes at IoU 0.50 (AP50), 0.75 (AP75) and the mean over 0.50:0.05:0.95 (AP), in percent."""
from typing import Dict, List, Sequence, Tuple

import numpy as np


def _iou_inclusive(a: Sequence[float], b: Sequence[float]) -> float:
    iw = min(a[2], b[2]) - max(a[0], b[0]) + 1.0
    ih = min(a[3], b[3]) - max(a[1], b[1]) + 1.0
    iw, ih = max(iw, 0.0), max(ih, 0.0)
    inter = iw * ih
    union = (a[2] - a[0] + 1.0) * (a[3] - a[1] + 1.0) + (b[2] - b[0] + 1.0) * (b[3] - b[1] + 1.0) - inter
    return inter / union


def _average_precision(recall: List[float], precision: List[float], eleven_point: bool) -> float:
    if eleven_point:
        total = 0.0
        # the thresholds are the devkit's `np.arange(0., 1.1, 0.1)` -- including its 0.30000000000000004 / 0.6000000000000001 /
        # 0.7000000000000001, which exclude a recall of exactly 3/10, 6/10, 7/10 where k / 10 would include it: the protocol
        # is what the toolkit computes (found by this oracle disagreeing with the product evaluator on its first draft)
        for t in (float(v) for v in np.arange(0.0, 1.1, 0.1)):
            best = 0.0
            for r, p in zip(recall, precision):
                if r >= t and p > best:
                    best = p
            total += best / 11.0
        return total
    r = [0.0] + list(recall) + [1.0]
    p = [0.0] + list(precision) + [0.0]
    for i in range(len(p) - 2, -1, -1):          # monotone envelope from the right
        if p[i + 1] > p[i]:
            p[i] = p[i + 1]
    area = 0.0
    for i in range(1, len(r)):
        if r[i] != r[i - 1]:
            area += (r[i] - r[i - 1]) * p[i]
    return area


def class_ap(dets: List[Tuple], gts: Dict[object, List[Tuple[Sequence[float], bool]]], thr: float, eleven_point: bool) -> float:
    """dets: [(image_id, score, x1, y1, x2, y2)] as read back from the text; gts: image -> [(box, difficult)]"""
    n_pos = sum(1 for boxes in gts.values() for _, diff in boxes if not diff)
    order = np.argsort(-np.array([d[1] for d in dets], dtype=np.float64)) if dets else []
    taken = {iid: [False] * len(boxes) for iid, boxes in gts.items()}
    tp, fp, recall, precision = 0, 0, [], []
    for k in order:
        iid, _, x1, y1, x2, y2 = dets[int(k)]
        cand = gts.get(iid, [])
        best, best_j = -float("inf"), -1
        for j, (box, _) in enumerate(cand):
            v = _iou_inclusive((x1, y1, x2, y2), box)
            if v > best:
                best, best_j = v, j
        if best > thr:
            if not cand[best_j][1]:
                if not taken[iid][best_j]:
                    taken[iid][best_j] = True
                    tp += 1
                else:
                    fp += 1
            # (a match with a difficult object counts as nothing)
        else:
            fp += 1
        recall.append(tp / n_pos if n_pos > 0 else 0.0)
        precision.append(tp / max(tp + fp, np.finfo(np.float64).eps))
    return _average_precision(recall, precision, eleven_point)


def evaluate(detections: List[Tuple], ground_truth: Dict[object, List[Tuple]], num_classes: int, is_2007: bool = False) -> Dict:
    """detections: [(image_id, class, score, x1, y1, x2, y2)] with 0-based corner boxes as the model emits them;
    ground_truth: image_id -> [(class, x1, y1, x2, y2, difficult)] with the 0-based boxes of the training records.
    Returns {"AP", "AP50", "AP75"} in percent."""
    per_thr = {}
    for t in range(50, 100, 5):
        aps = []
        for c in range(num_classes):
            dets = []
            for iid, cls, score, x1, y1, x2, y2 in detections:
                if cls != c:
                    continue
                line = f"{score:.3f} {x1 + 1:.1f} {y1 + 1:.1f} {x2:.1f} {y2:.1f}"
                s, a, b, cc, d = (float(v) for v in line.split(" "))
                dets.append((iid, s, a, b, cc, d))
            gts = {iid: [((x1 + 1.0, y1 + 1.0, x2, y2), bool(diff)) for cls, x1, y1, x2, y2, diff in rows if cls == c]
                   for iid, rows in ground_truth.items()}
            aps.append(100.0 * class_ap(dets, gts, t / 100.0, is_2007))
        per_thr[t] = float(np.mean(aps))
    return {"AP": float(np.mean(list(per_thr.values()))), "AP50": per_thr[50], "AP75": per_thr[75]}
