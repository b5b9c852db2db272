"""GPU parity at FUNCTION level on identical inputs (pytest -m gpu): the composed index-producing functions of the hot
path against the oracle's restatement of the same reference function, fed the SAME tensors --

  * `find_top_rpn_proposals`                 vs oracle.pt.find_top_rpn_proposals   (proposal_utils.py:27-154)
  * `GuassianFastRCNNOutputLayers.inference` vs oracle.pt.roi_inference            (fast_rcnn.py:34-141, 338-409)

Bar: counts, kept ROI indices, classes and ORDER exactly equal; clipped boxes bit-equal; scores 1e-5 (expf on the
device vs the host's vectorised exp differ in the last ulp).  Sizes are the BASELINE ones (37 350 anchors, 12 000 ->
2 000 proposals, 2 000 ROIs x 8 classes per image)."""
import numpy as np
import pytest
import torch

from oracle import d2, pt as opt
from tests.helpers import close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(K=8):
    from probabilisticteacher_amd.config import setup_cfg
    return setup_cfg("configs/pt/final_c2f.yaml", ["MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", "",
                                                   "MODEL.ROI_HEADS.NUM_CLASSES", K])


def _rand_boxes(gen, n, h, w, lo=8.0, hi=0.5):
    cx, cy = torch.rand(n, generator=gen) * w, torch.rand(n, generator=gen) * h
    bw = lo + torch.rand(n, generator=gen) * w * hi
    bh = lo + torch.rand(n, generator=gen) * h * hi
    return torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)


@pytest.mark.parametrize("n,r,pre,post,sizes", [
    (2, 37350, 12000, 2000, [(800, 1333), (800, 1303)]),        # BASELINE: 50 x 83 x 9 anchors, train / teacher top-k
    (3, 16650, 6000, 1000, [(600, 800), (576, 800), (600, 777)]),   # configs[0] map, test-time top-k
    (2, 300, 12000, 2000, [(64, 96), (64, 90)]),                # fewer anchors than pre_nms_topk
])
def test_find_top_rpn_proposals_identical_inputs(n, r, pre, post, sizes):
    from probabilisticteacher_amd.modeling.rpn import find_top_rpn_proposals
    gen = torch.Generator().manual_seed(r + n)
    H, W = sizes[0]
    decoded = _rand_boxes(gen, n * r, H, W, lo=-4.0, hi=0.6).view(n, r, 4)          # some degenerate (w <= 0) boxes
    decoded += torch.randn(n, r, 4, generator=gen) * 8                               # ... and some beyond the borders
    logits = torch.randn(n, r, generator=gen) * 3
    sigma = torch.randn(n, r, 4, generator=gen) * 2
    ocfg = opt.Cfg()
    ref = opt.find_top_rpn_proposals(ocfg, decoded, logits, sizes, sigma, pre, post, True)
    got = find_top_rpn_proposals(decoded.to(DEV), logits.to(DEV), sigma.to(DEV), sizes, ocfg.rpn_nms_thresh, pre, post,
                                 ocfg.rpn_min_box_size, True)
    assert len(got) == len(ref) == n
    for i, (a, b) in enumerate(zip(got, ref)):
        assert len(a) == len(b) and 0 < len(b) <= post, f"image {i}: {len(a)} vs {len(b)} proposals"
        assert torch.equal(a.proposal_boxes.tensor.cpu(), b.proposal_boxes.tensor), f"image {i}: proposal boxes / order"
        close(a.objectness_logits.cpu(), b.objectness_logits, 1e-5, 1e-6, "rescored logits")
    # non-finite predictions: training raises (proposal_utils.py:117-122), evaluation drops the rows
    bad = decoded.clone()
    bad[1, int(logits[1].argmax()), 2] = float("nan")          # the check only sees the top-k anchors (:117 after :92)
    with pytest.raises(FloatingPointError):
        find_top_rpn_proposals(bad.to(DEV), logits.to(DEV), sigma.to(DEV), sizes, 0.7, pre, post, 0.0, True)
    ref = opt.find_top_rpn_proposals(ocfg, bad, logits, sizes, sigma, pre, post, False)
    got = find_top_rpn_proposals(bad.to(DEV), logits.to(DEV), sigma.to(DEV), sizes, ocfg.rpn_nms_thresh, pre, post,
                                 ocfg.rpn_min_box_size, False)
    for a, b in zip(got, ref):
        assert torch.equal(a.proposal_boxes.tensor.cpu(), b.proposal_boxes.tensor)


def _roi_case(gen, K, counts, sizes, scale=2.0):
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    R = sum(counts)
    scores = torch.randn(R, K + 1, generator=gen) * scale
    deltas = torch.randn(R, 8 * K, generator=gen)
    deltas.view(R, K, 8)[..., :2] *= 2.0                   # centre shifts of a few tenths of the box
    deltas.view(R, K, 8)[..., 2:4] *= 1.5                  # size changes
    props_g, props_o = [], []
    for c, (h, w) in zip(counts, sizes):
        b = _rand_boxes(gen, c, h, w)
        b[:, 0::2].clamp_(0, w)
        b[:, 1::2].clamp_(0, h)
        pg, po = FreeInstances((h, w)), opt.FreeInstances((h, w))
        pg.proposal_boxes, po.proposal_boxes = Boxes(b.to(DEV)), d2.Boxes(b.clone())
        props_g.append(pg)
        props_o.append(po)
    return scores, deltas, props_g, props_o


def _check_inference(K, scores, deltas, props_g, props_o, ocfg):
    from probabilisticteacher_amd.modeling.roi_heads import GuassianFastRCNNOutputLayers
    layer = GuassianFastRCNNOutputLayers(_cfg(K), 1024)
    got, got_rows = layer.inference((scores.to(DEV), deltas.to(DEV)), props_g)
    ref, ref_rows = opt.roi_inference(ocfg, scores, deltas, props_o)
    assert len(got) == len(ref)
    for i, (a, b) in enumerate(zip(got, ref)):
        assert len(a) == len(b), f"image {i}: {len(a)} vs {len(b)} detections"
        assert torch.equal(got_rows[i].cpu(), ref_rows[i]), f"image {i}: kept ROI indices / order"
        assert torch.equal(a.pred_classes.cpu(), b.pred_classes), f"image {i}: classes"
        close(a.pred_boxes.tensor.cpu(), b.pred_boxes.tensor, 1e-5, 1e-4, "detection boxes")
        close(a.scores.cpu(), b.scores, 1e-5, 1e-7, "detection scores")
        assert torch.allclose(a.scores_logists.cpu(), b.scores_logists, rtol=0, atol=0, equal_nan=True), \
            "scores_logists are copies of input rows"
        assert torch.equal(a.boxes_sigma.cpu(), b.boxes_sigma), "boxes_sigma are copies of input entries"
    return ref


def test_roi_inference_identical_inputs_baseline_size():
    """teacher inference of a 2-image batch at BASELINE size: 2 000 proposals per image, K = 8 -> 16 000 (roi, class)
    candidates per image, per-class NMS through the fp32 offset trick, top 100"""
    K = 8
    gen = torch.Generator().manual_seed(7)
    ocfg = opt.Cfg(num_classes=K)
    scores, deltas, pg, po = _roi_case(gen, K, [2000, 1873], [(800, 1333), (800, 1200)])
    ref = _check_inference(K, scores, deltas, pg, po, ocfg)
    assert all(len(r) == 100 for r in ref)
    # peaked scores (a trained teacher): few candidates above the 0.05 threshold, fewer than 100 survive
    scores2 = scores.clone() * 0.2
    scores2[:, K] += 6.0
    idx = torch.randperm(scores2.shape[0], generator=gen)[:150]
    scores2[idx, torch.randint(0, K, (150,), generator=gen)] += 9.0
    ref = _check_inference(K, scores2, deltas, pg, po, ocfg)
    assert all(0 < len(r) < 100 for r in ref)


def test_roi_inference_edge_cases():
    """single-class head (K = 1, final_s2c), an image without any candidate, an image without proposals, and the finite
    filter with the reference's re-indexing quirk (fast_rcnn.py:68-71,96,126)"""
    gen = torch.Generator().manual_seed(11)
    ocfg1 = opt.Cfg(num_classes=1)
    scores, deltas, pg, po = _roi_case(gen, 1, [700, 300], [(600, 800), (512, 640)])
    scores[700:, 0] = -20.0                                # second image: nothing above the threshold
    ref = _check_inference(1, scores, deltas, pg, po, ocfg1)
    assert len(ref[0]) > 0 and len(ref[1]) == 0
    # an image without any proposal (the reference's `view(0, -1, 8)` would raise; here: no detections)
    from probabilisticteacher_amd.modeling.roi_heads import GuassianFastRCNNOutputLayers
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    empty = FreeInstances((600, 800))
    empty.proposal_boxes = Boxes(torch.zeros((0, 4), device=DEV))
    got, rows = GuassianFastRCNNOutputLayers(_cfg(1), 1024).inference((scores.to(DEV), deltas.to(DEV)),
                                                                      [pg[0], empty, pg[1]])
    assert [len(g) for g in got] == [len(ref[0]), 0, 0] and [len(r) for r in rows] == [len(ref[0]), 0, 0]
    close(got[0].pred_boxes.tensor.cpu(), ref[0].pred_boxes.tensor, 1e-5, 1e-4, "detections next to an empty image")
    K = 8
    ocfg = opt.Cfg(num_classes=K)
    scores, deltas, pg, po = _roi_case(gen, K, [400, 350], [(600, 800), (600, 800)])
    deltas[3, 2] = float("inf")                            # image 0, ROI 3: one decoded box is not finite
    deltas[17, 9] = float("nan")
    scores[420, 2] = float("nan")                          # image 1, ROI 20: probabilities are NaN
    _check_inference(K, scores, deltas, pg, po, ocfg)
