"""CPU: host-side logic of the plug-in layer (config, registry, structures, LR schedule, flat buffers, sampler)."""
import math
import os

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


def test_checkpoint_layout_roundtrip(tmp_path):
    """modelTeacher.* / modelStudent.* key layout (ts_ensemble.py:20-29) survives a save -> load round trip."""
    import types
    from probabilisticteacher_amd import checkpoint
    from probabilisticteacher_amd.engine.flat import FlatParams
    from probabilisticteacher_amd.modeling import EnsembleTSModel
    torch.manual_seed(0)
    s, t = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3)
    tr = types.SimpleNamespace(model=s, model_teacher=t, ensem_ts_model=EnsembleTSModel(t, s), iter=7, start_iter=0,
                               student=FlatParams(s), _first_step=False)
    tr.momentum_buf = torch.arange(tr.student.n_trainable, dtype=torch.float32)
    p = str(tmp_path / "model_0000006.pth")
    checkpoint.save_checkpoint(tr, p)
    raw = torch.load(p)
    assert sorted(raw["model"]) == ["modelStudent.bias", "modelStudent.weight", "modelTeacher.bias", "modelTeacher.weight"]
    want = {k: v.clone() for k, v in tr.ensem_ts_model.state_dict().items()}
    with torch.no_grad():
        for v in tr.ensem_ts_model.state_dict().values():
            v.zero_()
    tr.iter, tr.momentum_buf = 0, torch.zeros_like(tr.momentum_buf)
    checkpoint.load_checkpoint(tr, p)
    assert all(torch.equal(v, want[k]) for k, v in tr.ensem_ts_model.state_dict().items())
    assert tr.iter == 7 and float(tr.momentum_buf[-1]) == tr.student.n_trainable - 1


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
