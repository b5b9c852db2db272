"""CPU: host-side logic of the plug-in layer (config, registry, structures, LR schedule, flat buffers, sampler)."""
import itertools
import math
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_surface_matches_reference_keys():
    from probabilisticteacher_amd.config import setup_cfg
    cfg = setup_cfg(os.path.join(ROOT, "configs/pt/final_c2f.yaml"),
                    ["MODEL.ANCHOR_GENERATOR.NAME", "DifferentiableAnchorGenerator", "UNSUPNET.TAU", "[0.5,0.5]"])
    assert cfg.MODEL.META_ARCHITECTURE == "GuassianGeneralizedRCNN"
    assert cfg.MODEL.BACKBONE.NAME == "build_vgg_backbone" and cfg.MODEL.PROPOSAL_GENERATOR.NAME == "GuassianRPN"
    assert cfg.MODEL.RPN.HEAD_NAME == "GuassianRPNHead" and cfg.MODEL.ROI_HEADS.NAME == "GuassianROIHead"
    assert cfg.MODEL.ROI_HEADS.NUM_CLASSES == 8 and cfg.MODEL.RPN.POSITIVE_FRACTION == 0.25
    assert cfg.UNSUPNET.TAU == [0.5, 0.5] and cfg.UNSUPNET.BURN_UP_STEP == 4000 and cfg.UNSUPNET.EMA_KEEP_RATE == 0.9996
    assert cfg.SOLVER.STEPS == (30000,) and cfg.SOLVER.BASE_LR == 0.016 and cfg.SOLVER.WARMUP_ITERS == 400
    assert len(cfg.MODEL.ANCHOR_GENERATOR.ANCHOR[0]) == 9
    assert cfg.is_frozen()
    with pytest.raises(AttributeError):
        cfg.SOLVER.BASE_LR = 1.0
    with pytest.raises(KeyError):
        setup_cfg("", ["MODEL.NOPE", 1])
    s2c = setup_cfg(os.path.join(ROOT, "configs/pt/final_s2c.yaml"))
    assert s2c.MODEL.ROI_HEADS.NUM_CLASSES == 1


def test_registries_and_state_dict_names():
    from probabilisticteacher_amd import modeling  # noqa: F401  (registers)
    from probabilisticteacher_amd import registry as R
    from probabilisticteacher_amd.config import setup_cfg
    for reg, name in ((R.META_ARCH_REGISTRY, "GuassianGeneralizedRCNN"), (R.BACKBONE_REGISTRY, "build_vgg_backbone"),
                      (R.PROPOSAL_GENERATOR_REGISTRY, "GuassianRPN"), (R.RPN_HEAD_REGISTRY, "GuassianRPNHead"),
                      (R.ROI_HEADS_REGISTRY, "GuassianROIHead"), (R.ANCHOR_GENERATOR_REGISTRY, "DifferentiableAnchorGenerator"),
                      (R.ANCHOR_GENERATOR_REGISTRY, "DefaultAnchorGenerator"), (R.ROI_BOX_HEAD_REGISTRY, "FastRCNNConvFCHead")):
        assert name in reg
    with pytest.raises(KeyError):
        R.META_ARCH_REGISTRY.get("nope")
    cfg = setup_cfg(os.path.join(ROOT, "configs/pt/final_c2f.yaml"),
                    ["MODEL.DEVICE", "cpu", "MODEL.VGG.PRETRAIN", "", "MODEL.ANCHOR_GENERATOR.NAME", "DifferentiableAnchorGenerator"])
    model = modeling.build_model(cfg)
    from oracle import pt as opt
    ref = opt.init_params(opt.Cfg(anchor_generator="DifferentiableAnchorGenerator"), 0)
    sd = model.state_dict()
    assert set(sd) == set(ref)
    assert all(sd[k].shape == ref[k].shape for k in ref)
    frozen = sum(p.numel() for p in model.parameters() if not p.requires_grad)
    assert frozen == 260160                                          # blocks 1-2
    assert sum(v.numel() for v in sd.values()) == 43931610 + 18      # + the (9,2) anchor table
    assert torch.allclose(sd["proposal_generator.anchor_generator.anchor_0"],
                          ref["proposal_generator.anchor_generator.anchor_0"], atol=1e-3)


def test_lr_schedule_and_flat_buffers():
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine.flat import FlatParams, lr_at
    from oracle import d2
    cfg = setup_cfg(os.path.join(ROOT, "configs/pt/final_c2f.yaml"))
    for it in (0, 1, 399, 400, 29999, 30000):
        assert math.isclose(lr_at(cfg, it), d2.warmup_multistep_lr(it, 0.016, (30000,), 0.1, 1e-3, 400), rel_tol=1e-12)
    assert math.isclose(lr_at(cfg, 0), 0.016e-3) and math.isclose(lr_at(cfg, 400), 0.016) and math.isclose(lr_at(cfg, 30000), 0.0016)
    m = torch.nn.Linear(4, 3)
    w0 = m.weight.detach().clone()
    flat = FlatParams(m)
    assert torch.equal(m.weight, w0) and m.weight.data_ptr() == flat.flat.data_ptr()
    flat.zero_grad()
    m(torch.ones(2, 4)).sum().backward()
    assert flat.grad.abs().sum() > 0 and m.weight.grad.data_ptr() == flat.grad.data_ptr()
    flat.flat.zero_()
    assert float(m.weight.abs().sum()) == 0.0                        # parameters are views of the flat buffer


def test_structures_and_sampler():
    from probabilisticteacher_amd.modeling import sampling
    from probabilisticteacher_amd.structures import Boxes, FreeInstances, Instances
    from oracle import d2, pt as opt
    b = Boxes(torch.tensor([[-5.0, 2.0, 30.0, 50.0], [3.0, 3.0, 3.0, 9.0]]))
    b.clip((40, 20))
    assert b.tensor.tolist() == [[0.0, 2.0, 20.0, 40.0], [3.0, 3.0, 3.0, 9.0]]
    assert b.nonempty().tolist() == [True, False]
    inst = Instances((10, 10), a=torch.arange(3))
    with pytest.raises(AssertionError):
        inst.b = torch.arange(4)
    f = FreeInstances((10, 10), a=torch.arange(3))
    f.b = torch.arange(4)                                            # no equal-length check (instances.py:27)
    assert len(f) == 3 and isinstance(f.to("cpu"), FreeInstances)
    # the keyed sampler driven by keys that encode a permutation sequence == D2 subsample_labels with those permutations
    from tests.helpers import perm_key_source
    labels = torch.tensor([[1, 0, -1, 0, 1, 1, 0, 0, 0, 1]], dtype=torch.int8)
    sampling.set_key_source(perm_key_source(opt.SeededPerm(3)))
    try:
        got = sampling.keyed_relabel(labels, 4, 0.5, 0)
    finally:
        sampling.set_key_source(None)
    rp, rn = d2.subsample_labels(labels[0], 4, 0.5, 0, opt.SeededPerm(3))
    ref = torch.full_like(labels[0], -1)
    ref[rp], ref[rn] = 1, 0
    assert torch.equal(got[0], ref) and len(rp) == 2 and len(rn) == 2




def test_detector_postprocess_rescales_clips_and_drops_empty():
    """D2 detector_postprocess as used by the eval path: scale to the requested output size, clip,
    drop boxes that become empty."""
    import torch
    from probabilisticteacher_amd.modeling.meta_arch import detector_postprocess
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    r = FreeInstances((100, 200))
    r.pred_boxes = Boxes(torch.tensor([[10.0, 20.0, 50.0, 60.0], [190.0, 90.0, 260.0, 140.0], [300.0, 10.0, 320.0, 30.0]]))
    r.scores = torch.tensor([0.9, 0.8, 0.7])
    out = detector_postprocess(r, 50, 400)          # x * 2, y * 0.5
    assert out.image_size == (50, 400)
    assert torch.equal(out.pred_boxes.tensor, torch.tensor([[20.0, 10.0, 100.0, 30.0], [380.0, 45.0, 400.0, 50.0]]))
    assert torch.equal(out.scores, torch.tensor([0.9, 0.8]))
    assert torch.equal(r.pred_boxes.tensor[0], torch.tensor([10.0, 20.0, 50.0, 60.0]))      # input not mutated


def test_keyed_sampling_equals_reference_subsample_labels():
    """The sync-free sampler (random keys + top-k) selects exactly what D2's subsample_labels selects when the latter
    draws its two permutations as the key order of the candidates (oracle KeyedPerm) -- labels, counts and order."""
    import torch
    from oracle import d2, pt as opt
    from probabilisticteacher_amd.modeling import sampling
    g = torch.Generator().manual_seed(5)
    # --- RPN flavour: batch relabel, 256 samples @ 0.5, bg label 0; rows with few / no positives and few negatives
    n, r = 4, 3000
    labels = torch.full((n, r), -1, dtype=torch.int8)
    labels[0, torch.randperm(r, generator=g)[:400]] = 1
    labels[0, torch.randperm(r, generator=g)[:2000]] = 0
    labels[1, torch.randperm(r, generator=g)[:20]] = 1           # fewer positives than 128 -> more negatives
    labels[1, torch.randperm(r, generator=g)[:1500]] = 0
    labels[2, torch.randperm(r, generator=g)[:50]] = 0            # no positives, fewer negatives than 256
    labels[3, :] = 1                                              # no negatives at all
    kp = opt.KeyedPerm(77)
    from tests.helpers import keyed_perm_source
    sampling.set_key_source(keyed_perm_source(kp))
    try:
        got = sampling.keyed_relabel(labels.clone(), 256, 0.5, 0)
    finally:
        sampling.set_key_source(None)
    kp.start_replay()
    for i in range(n):
        lab = labels[i].clone()
        pos, neg = d2.subsample_labels(lab, 256, 0.5, 0, kp)
        ref = torch.full_like(lab, -1)
        ref[pos] = 1
        ref[neg] = 0
        assert torch.equal(got[i], ref), f"row {i}"
    p1 = int((labels[1] == 1).sum())
    assert 0 < p1 <= 20 and int((got[1] == 1).sum()) == p1 and int((got[1] == 0).sum()) == 256 - p1
    assert int((got[2] == 1).sum()) == 0 and int((got[2] == 0).sum()) == 50 and int((got[3] == 1).sum()) == 128


def test_lr_schedules_match_the_reference_table():
    """WarmupTwoStageMultiStepLR: lr(it) of the REAL reference class driving a real torch SGD (tests/golden/
    solver_checkpoint.npz, tools/gen_golden.py::gen_solver_checkpoint); WarmupCosineLR / unknown names; optimiser options
    that the fused step does not implement fail loudly."""
    import numpy as np
    import pytest
    from probabilisticteacher_amd import solver
    from probabilisticteacher_amd.config import setup_cfg
    z = np.load(os.path.join(ROOT, "tests", "golden", "solver_checkpoint.npz"))
    base, wf, wi, s0, s1, f0, f1, f2 = [float(v) for v in z["lr_twostage_cfg"]]
    cfg = setup_cfg(os.path.join(ROOT, "configs/pt/final_c2f.yaml"), [
        "SOLVER.LR_SCHEDULER_NAME", "WarmupTwoStageMultiStepLR", "SOLVER.BASE_LR", base, "SOLVER.WARMUP_FACTOR", wf,
        "SOLVER.WARMUP_ITERS", int(wi), "SOLVER.STEPS", (int(s0), int(s1)), "SOLVER.FACTOR_LIST", (f0, f1, f2)])
    got = [solver.lr_at(cfg, it) for it in range(len(z["lr_twostage"]))]
    assert np.allclose(got, z["lr_twostage"], rtol=1e-12, atol=0)
    cos = setup_cfg(os.path.join(ROOT, "configs/pt/final_c2f.yaml"), ["SOLVER.LR_SCHEDULER_NAME", "WarmupCosineLR",
                                                                    "SOLVER.MAX_ITER", 1000, "SOLVER.WARMUP_ITERS", 10])
    assert math.isclose(solver.lr_at(cos, 500), 0.016 * 0.5) and math.isclose(solver.lr_at(cos, 0), 0.016 * 1e-3)
    for bad in (["SOLVER.LR_SCHEDULER_NAME", "StepLR"], ["SOLVER.NESTEROV", True], ["SOLVER.BIAS_LR_FACTOR", 2.0],
                ["SOLVER.WEIGHT_DECAY_BIAS", 0.0], ["SOLVER.STEPS", (5, 3), "SOLVER.LR_SCHEDULER_NAME", "WarmupTwoStageMultiStepLR",
                                                   "SOLVER.FACTOR_LIST", (1, 1, 1)]):
        with pytest.raises(ValueError):
            solver.check_optimizer_options(setup_cfg(os.path.join(ROOT, "configs/pt/final_c2f.yaml"), bad))


def _cpu_trainer(tmp_path, extra=()):
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer
    cfg = setup_cfg(os.path.join(ROOT, "configs/pt/final_c2f.yaml"), [
        "MODEL.DEVICE", "cpu", "MODEL.VGG.PRETRAIN", "", "MODEL.ANCHOR_GENERATOR.NAME", "DifferentiableAnchorGenerator",
        "OUTPUT_DIR", str(tmp_path)] + list(extra))
    return cfg, PTrainer(cfg)


def test_checkpoint_file_has_the_reference_layout(tmp_path):
    """The file `save_checkpoint` writes against what the reference's own classes produce (EnsembleTSModel key layout,
    D2's per-parameter SGD groups in modules() order, torch SGD / scheduler state_dict keys, fvcore's top-level dict;
    tests/golden/solver_checkpoint.npz), and the off-by-one-free iteration bookkeeping: "iteration" = the
    iteration that just finished, resume continues at iteration + 1."""
    import numpy as np
    from probabilisticteacher_amd import checkpoint
    z = np.load(os.path.join(ROOT, "tests", "golden", "solver_checkpoint.npz"))
    cfg, tr = _cpu_trainer(tmp_path)
    g = torch.Generator().manual_seed(1)
    tr.momentum_buf.copy_(torch.randn(tr.momentum_buf.shape, generator=g))
    tr._first_step = False
    tr.iter = 2                                               # two iterations (0 and 1) are done
    path = checkpoint.PeriodicCheckpointer(tr, period=2, max_iter=100).step(1)
    assert os.path.basename(path) == "model_0000001.pth" and checkpoint.last_checkpoint(str(tmp_path)) == path
    raw = torch.load(path, weights_only=False)
    assert list(raw.keys()) == list(z["top_keys"]) and raw["iteration"] == int(z["iteration"][0]) == 1
    assert list(raw["model"].keys()) == list(z["model_keys"])
    assert [",".join(str(d) for d in v.shape) for v in raw["model"].values()] == list(z["model_shapes"])
    osd = raw["optimizer"]
    assert [n for n, p in tr.model.named_parameters() if p.requires_grad] == list(z["opt_index_names"])
    assert [gr["params"][0] for gr in osd["param_groups"]] == list(z["opt_group_params"])
    assert set(osd["param_groups"][0]) <= set(z["opt_group_keys"]) and {"lr", "momentum", "weight_decay", "nesterov", "params"} <= set(osd["param_groups"][0])
    assert sorted(osd["state"][0].keys()) == list(z["opt_state_keys"])
    assert raw["scheduler"]["last_epoch"] == int(z["sched_last_epoch"][0]) == 2 and set(raw["scheduler"]) <= set(z["sched_keys"])
    # a real torch.optim.SGD (the reference's optimiser class) built the D2 way accepts the state, and its momentum
    # buffers are the slices of the flat buffer
    params = [p for _, p in tr.model.named_parameters() if p.requires_grad]
    sgd = torch.optim.SGD([{"params": [p]} for p in params], lr=0.016, momentum=0.9)
    sgd.load_state_dict(osd)
    names = list(z["opt_index_names"])
    for j in (0, 5, len(names) - 1):
        off, k = tr.student.index[names[j]]
        assert torch.equal(sgd.state[params[j]]["momentum_buffer"].reshape(-1), tr.momentum_buf[off:off + k])
    # ... and the way back: a state_dict produced by torch SGD loads into the flat buffer; resume = iteration + 1
    cfg2, tr2 = _cpu_trainer(tmp_path)
    raw["optimizer"] = sgd.state_dict()
    torch.save(raw, path)
    inc = checkpoint.load_checkpoint(tr2, path, resume=True)
    assert inc == checkpoint.IncompatibleKeys([], [], []) and tr2.iter == tr2.start_iter == 2 and not tr2._first_step
    assert torch.equal(tr2.momentum_buf, tr.momentum_buf) and torch.equal(tr2.student.flat, tr.student.flat)
    assert torch.equal(tr2.teacher.flat, tr.teacher.flat)
    # weights only (resume=False): iteration and momentum untouched
    cfg3, tr3 = _cpu_trainer(tmp_path)
    checkpoint.load_checkpoint(tr3, path, resume=False)
    assert tr3.iter == 0 and tr3._first_step and float(tr3.momentum_buf.abs().sum()) == 0.0
    assert torch.equal(tr3.student.flat, tr.student.flat)


def test_student_only_checkpoints_prefix_shapes_and_final(tmp_path):
    """detection_checkpoint.py:26-50,75-103: a Caffe2-tagged / bare-key file updates the student only; a `module.` prefix
    is stripped; wrongly-shaped tensors are reported and skipped (never broadcast); missing keys are reported;
    PeriodicCheckpointer writes model_final.pth after the last iteration."""
    from probabilisticteacher_amd import checkpoint
    cfg, tr = _cpu_trainer(tmp_path)
    t0 = tr.teacher.flat.clone()
    sd = {("module." + k): torch.full_like(v, 0.5) for k, v in tr.model.state_dict().items()}
    k_bad = "module.roi_heads.box_predictor.cls_score.bias"
    sd[k_bad] = torch.zeros(1)                                # would broadcast under a bare copy_()
    del sd["module.proposal_generator.rpn_head.conv.bias"]
    sd["module.not_in_model"] = torch.zeros(2)
    p = str(tmp_path / "student.pth")
    torch.save({"model": sd, "__author__": "Caffe2"}, p)
    before = tr.model.state_dict()["roi_heads.box_predictor.cls_score.bias"].clone()
    inc = checkpoint.load_checkpoint(tr, p, resume=False)
    assert inc.missing_keys == ["proposal_generator.rpn_head.conv.bias"] and inc.unexpected_keys == ["not_in_model"]
    assert inc.incorrect_shapes == [("roi_heads.box_predictor.cls_score.bias", (1,), tuple(before.shape))]
    ssd = tr.model.state_dict()
    assert torch.equal(ssd["roi_heads.box_predictor.cls_score.bias"], before)
    assert float(ssd["backbone.vgg_block4.0.conv2.weight"].mean()) == 0.5 and torch.equal(tr.teacher.flat, t0)
    tr.iter = 10
    out = checkpoint.PeriodicCheckpointer(tr, period=0, max_iter=10).step(9)
    assert os.path.basename(out) == "model_final.pth" and torch.load(out, weights_only=False)["iteration"] == 9


def test_vgg_pretrained_key_map_matches_the_reference(tmp_path):
    """vgg.py:127-152: which `features.N` tensor of vgg16_caffe.pth each backbone parameter receives, as recorded from the
    real VGG.__init__ (every source tensor carried a distinct constant), and which parameters FREEZE_AT=2 freezes."""
    import numpy as np
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.modeling import build_model
    z = np.load(os.path.join(ROOT, "tests", "golden", "solver_checkpoint.npz"))
    cfg0 = setup_cfg(os.path.join(ROOT, "configs/pt/final_c2f.yaml"), ["MODEL.DEVICE", "cpu", "MODEL.VGG.PRETRAIN", ""])
    shapes = {k: v.shape for k, v in build_model(cfg0).backbone.state_dict().items()}
    sd, val = {}, {}
    for mk, sk in zip(z["vgg_model_keys"], z["vgg_source_keys"]):
        val[str(sk)] = float(len(val) + 1)
        sd[str(sk)] = torch.full(shapes[str(mk)], val[str(sk)])
    path = str(tmp_path / "vgg16_caffe.pth")
    torch.save(sd, path)
    cfg = setup_cfg(os.path.join(ROOT, "configs/pt/final_c2f.yaml"), ["MODEL.DEVICE", "cpu", "MODEL.VGG.PRETRAIN", path])
    bb = build_model(cfg).backbone
    got = bb.state_dict()
    assert list(got.keys()) == list(z["vgg_model_keys"])
    for mk, sk in zip(z["vgg_model_keys"], z["vgg_source_keys"]):
        assert float(got[str(mk)].flatten()[0]) == val[str(sk)], f"{mk} <- {sk}"
    assert [n for n, p in bb.named_parameters() if not p.requires_grad] == list(z["vgg_frozen"])


def test_voc_evaluator_known_answers():
    """VOC protocol (D2 PascalVOCDetectionEvaluator as wired at reference trainer.py:127-137): hand-computed cases --
    greedy matching in score order, duplicate detections are false positives, 'difficult' objects are ignored, the +1
    inclusive IoU, 1-based text round trip, area-rule AP vs the 11-point rule, AP50 / AP75 / AP averaging."""
    import numpy as np
    from probabilisticteacher_amd.evaluation import PascalVOCDetectionEvaluator, voc_ap, voc_eval
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    assert abs(voc_ap(np.array([0.5, 1.0]), np.array([1.0, 2 / 3])) - (0.5 * 1.0 + 0.5 * 2 / 3)) < 1e-12
    assert abs(voc_ap(np.array([0.5, 1.0]), np.array([1.0, 2 / 3]), True) - (6 * 1.0 + 5 * 2 / 3) / 11) < 1e-12
    gts = {0: {"bbox": [[11, 11, 50, 50], [61, 61, 100, 100]], "difficult": [False, True]}, 1: {"bbox": [[1, 1, 20, 20]], "difficult": [False]}}
    dets = [(0, 0.9, 11, 11, 50, 50), (0, 0.8, 12, 12, 50, 50), (0, 0.7, 61, 61, 100, 100), (1, 0.6, 100, 100, 120, 120), (1, 0.5, 1, 1, 20, 21)]
    rec, prec, ap = voc_eval(dets, gts, 0.5)
    # tp fp(dup) ignored(difficult) fp tp ; npos = 2
    assert rec.tolist() == [0.5, 0.5, 0.5, 0.5, 1.0] and np.allclose(prec, [1, 0.5, 0.5, 1 / 3, 0.5])
    assert abs(ap - (0.5 * 1.0 + 0.5 * 0.5)) < 1e-12
    # inclusive IoU: boxes [1,1,10,10] vs [1,1,10,20] -> 100 / 200 = 0.5 exactly: NOT a match at threshold 0.5 (strict >)
    assert voc_eval([(0, 1.0, 1, 1, 10, 20)], {0: {"bbox": [[1, 1, 10, 10]], "difficult": [False]}}, 0.5)[2] == 0.0
    ev = PascalVOCDetectionEvaluator(["car", "person"])
    gt = FreeInstances((100, 100))
    gt.gt_boxes, gt.gt_classes = Boxes(torch.tensor([[10.0, 10.0, 50.0, 50.0], [60.0, 20.0, 90.0, 80.0]])), torch.tensor([0, 1])
    det = FreeInstances((100, 100))
    det.pred_boxes = Boxes(torch.tensor([[10.0, 10.0, 50.0, 50.0], [60.0, 20.0, 90.0, 70.0], [0.0, 0.0, 5.0, 5.0]]))
    det.scores, det.pred_classes = torch.tensor([0.9, 0.8, 0.3]), torch.tensor([0, 1, 1])
    ev.process([{"image_id": 7, "instances": gt}], [{"instances": det}])
    res = ev.evaluate()
    # car: perfect at every threshold (AP 100).  person: IoU = (31*51)/(31*61) = 0.836 -> hit for thresholds <= 0.80
    assert res["per_class_AP50"] == {"car": 100.0, "person": 100.0}
    assert abs(res["bbox"]["AP50"] - 100.0) < 1e-9 and abs(res["bbox"]["AP75"] - 100.0) < 1e-9
    assert abs(res["bbox"]["AP"] - (100.0 * 10 + 100.0 * 7) / 20) < 1e-9


def _write_voc_dir(root, ids, class_names, rng, h=96, w=128, with_difficult=True):
    """a tiny VOC-format dataset: JPEGImages/<id>.jpg, Annotations/<id>.xml, ImageSets/Main/train.txt"""
    import os
    from PIL import Image
    for sub in ("JPEGImages", "Annotations", "ImageSets/Main"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    truth = {}
    for fid in ids:
        img = rng.randint(0, 96, (h, w, 3)).astype(np.uint8)
        m = int(rng.randint(1, 4))
        objs = []
        for _ in range(m):
            bw, bh = int(rng.randint(24, 60)), int(rng.randint(20, 50))
            x1, y1 = int(rng.randint(1, w - bw)), int(rng.randint(1, h - bh))
            img[y1:y1 + bh, x1:x1 + bw] += 120
            objs.append((class_names[int(rng.randint(len(class_names)))], x1, y1, x1 + bw, y1 + bh, int(with_difficult and rng.rand() < 0.25)))
        objs.append(("ignored-class", 1, 1, 9, 9, 0))
        Image.fromarray(img).save(os.path.join(root, "JPEGImages", fid + ".jpg"), quality=95)
        xml = f"<annotation><size><width>{w}</width><height>{h}</height><depth>3</depth></size>" + "".join(
            f"<object><name>{n}</name><difficult>{d}</difficult><bndbox><xmin>{a}</xmin><ymin>{b}</ymin><xmax>{c}</xmax><ymax>{e}</ymax>"
            f"</bndbox></object>" for n, a, b, c, e, d in objs) + "</annotation>"
        open(os.path.join(root, "Annotations", fid + ".xml"), "w").write(xml)
        truth[fid] = objs[:-1]
    open(os.path.join(root, "ImageSets", "Main", "train.txt"), "w").write("\n".join(ids) + "\n")
    return truth


def test_voc_directory_reader_and_training_sampler(tmp_path):
    """pt/data/datasets/builtin.py + D2 load_voc_instances / TrainingSampler restated in probabilisticteacher_amd/data"""
    from probabilisticteacher_amd.data import datasets, training_sampler
    rng = np.random.RandomState(1)
    names = ("car", "person")
    truth = _write_voc_dir(str(tmp_path), ["a1", "b2", "c3"], names, rng)
    datasets.register_pascal_voc("tiny_train", str(tmp_path), "train", names)
    dicts = datasets.get_dataset_dicts(["tiny_train"], filter_empty=True)
    assert [d["image_id"] for d in dicts] == ["a1", "b2", "c3"] and datasets.metadata("tiny_train")["thing_classes"] == list(names)
    for d in dicts:
        assert (d["height"], d["width"]) == (96, 128) and len(d["annotations"]) == len(truth[d["image_id"]])
        for a, (n, x1, y1, x2, y2, diff) in zip(d["annotations"], truth[d["image_id"]]):
            assert a["bbox"] == [x1 - 1.0, y1 - 1.0, float(x2), float(y2)] and a["category_id"] == names.index(n) and a["difficult"] == diff
        m = datasets.to_mapper_input(d)
        assert m["image"].dtype == torch.uint8 and tuple(m["image"].shape) == (3, 96, 128) and m["boxes"].shape == (len(d["annotations"]), 4)
    with pytest.raises(KeyError):
        datasets.get_dataset_dicts(["nope"])
    a = list(itertools.islice(training_sampler(5, seed=3), 12))
    b = list(itertools.islice(training_sampler(5, seed=3), 12))
    assert a == b and sorted(a[:5]) == [0, 1, 2, 3, 4] and sorted(a[5:10]) == [0, 1, 2, 3, 4], "seeded epochs of a full permutation"


def test_voc_evaluator_against_the_independent_oracle():
    """probabilisticteacher_amd/evaluation.py against oracle/voc.py (written separately: plain-Python loops, no shared code) on
    241 detections, 14 images, 3 classes, difficult objects, duplicate and cross-class detections, score ties after the
    3-decimal text round trip -- AP / AP50 / AP75 to 1e-9, VOC2010+ and VOC2007 rules."""
    from oracle import voc
    from probabilisticteacher_amd.evaluation import PascalVOCDetectionEvaluator
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    rng = np.random.RandomState(0)
    K = 3
    for is07 in (False, True):
        ev = PascalVOCDetectionEvaluator([f"c{i}" for i in range(K)], is_2007=is07)
        dets, gt = [], {}
        for iid in range(14):
            m = int(rng.randint(0, 5))
            b = rng.rand(m, 2) * 60
            boxes = np.concatenate([b, b + 10 + rng.rand(m, 2) * 40], 1).astype(np.float32)
            cls, diff = rng.randint(0, K, m), rng.rand(m) < 0.2
            inst = FreeInstances((128, 128))
            inst.gt_boxes, inst.gt_classes, inst.difficult = Boxes(torch.from_numpy(boxes)), torch.from_numpy(cls), torch.from_numpy(diff)
            gt[iid] = [(int(c), *[float(v) for v in bb], bool(d)) for c, bb, d in zip(cls, boxes, diff)]
            db, dc, ds = [], [], []
            for _ in range(int(rng.randint(8, 28))):
                if m and rng.rand() < 0.7:
                    j = int(rng.randint(m))
                    bb, c = boxes[j] + rng.randn(4) * 3, (cls[j] if rng.rand() < 0.85 else rng.randint(K))
                else:
                    q = rng.rand(2) * 60
                    bb, c = np.concatenate([q, q + 10 + rng.rand(2) * 40]), rng.randint(K)
                db.append(bb), dc.append(int(c)), ds.append(round(float(rng.rand()), 2))
            out = FreeInstances((128, 128))
            out.pred_boxes, out.scores = Boxes(torch.tensor(np.array(db), dtype=torch.float32)), torch.tensor(ds, dtype=torch.float32)
            out.pred_classes = torch.tensor(dc)
            ev.process([{"image_id": iid, "instances": inst}], [{"instances": out}])
            dets += [(iid, c, float(s), *[float(v) for v in bb]) for bb, c, s in zip(out.pred_boxes.tensor.numpy(), dc, out.scores.tolist())]
        assert len(dets) >= 200
        got, ref = ev.evaluate()["bbox"], voc.evaluate(dets, gt, K, is_2007=is07)
        for k in ("AP", "AP50", "AP75"):
            assert 5.0 < ref[k] < 95.0 and abs(got[k] - ref[k]) < 1e-9, (k, got, ref)


def test_eval_hooks_flatten_student_and_teacher_results():
    """reference trainer.py:529-542: student results suffixed `_student`, teacher results plain, flattened like D2's EvalHook"""
    from types import SimpleNamespace
    from probabilisticteacher_amd.engine.trainer import PTrainer
    me = SimpleNamespace(cfg=None, model="S", model_teacher="T")
    calls = []

    def fake_test(cfg, model):
        calls.append(model)
        return {"bbox": {"AP": 10.0 if model == "S" else 20.0, "AP50": 30.0, "AP75": 5.0}, "per_class_AP50": {"car": 1.0}}
    flat = PTrainer._run_eval_hooks(me, fake_test)
    assert calls == ["S", "T"], "student first, then teacher"
    assert flat["bbox_student/AP"] == 10.0 and flat["bbox/AP"] == 20.0 and flat["bbox_student/AP50"] == 30.0
    assert flat["per_class_AP50_student/car"] == 1.0 and me._last_eval_results_teacher["bbox"]["AP"] == 20.0

