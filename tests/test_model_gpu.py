"""GPU parity (pytest -m gpu): the HIP model / trainer against (a) golden vectors produced by the REAL reference
(tests/golden, see tools/gen_golden.py) and (b) the CPU oracle on the same seeded inputs.

Tolerances: losses rtol 1e-4 (BASELINE.json north_star: "loss parity vs CPU reference to 1e-4"); gradients
rtol 2e-3 of the tensor norm (fp32 MFMA k-ordered accumulation vs oneDNN blocking); box coordinates 2e-3 px;
class indices, sampler draw sizes (== label counts) exact."""
import math

import numpy as np
import pytest
import torch

from oracle import d2, pt as opt
from tests.helpers import close, keyed_perm_source, load, match_detections, perm_key_source, records

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(K, anchor, tau, burn=4000):
    from probabilisticteacher_amd.config import setup_cfg
    return setup_cfg("configs/pt/final_c2f.yaml", [
        "MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", "", "MODEL.ANCHOR_GENERATOR.NAME", anchor,
        "MODEL.ROI_HEADS.NUM_CLASSES", K, "UNSUPNET.TAU", list(tau), "UNSUPNET.BURN_UP_STEP", burn])


def _load_params(model, params):
    sd = model.state_dict()
    assert set(sd) == set(params), set(sd) ^ set(params)
    with torch.no_grad():
        for k, v in params.items():
            sd[k].copy_(v)


def _gpu_records(z, prefix, n):
    from probabilisticteacher_amd.structures import FreeInstances
    return records(z, prefix, n, make_instances=FreeInstances)


def _grad_check(z, prefix, named, tol=2e-3):
    checked = 0
    for k, p in named.items():
        key = f"{prefix}_norm_{k}"
        if key not in z.files:
            continue
        g = p.grad
        nrm = float(z[key])
        close(g.double().norm().cpu(), z[key], tol, 1e-7, key)
        close(g.flatten()[:32].cpu(), z[f"{prefix}_head_{k}"], 5e-3, 2e-4 * nrm + 1e-8, f"{prefix}_head_{k}")
        checked += 1
    assert checked >= 4


@pytest.mark.parametrize("anchor,tag", [("DefaultAnchorGenerator", "default_anchor"),
                                        ("DifferentiableAnchorGenerator", "diff_anchor")])
def test_model_branches_match_reference_goldens(anchor, tag):
    from probabilisticteacher_amd import modeling
    from probabilisticteacher_amd.modeling import sampling
    from probabilisticteacher_amd.engine.flat import FlatParams
    z = load("model_" + tag)
    K, tau = int(z["K"]), tuple(float(v) for v in z["tau"])
    cfg = _cfg(K, anchor, tau)
    ocfg = opt.Cfg(num_classes=K, anchor_generator=anchor, tau=tau)
    model = modeling.build_model(cfg)
    model.train()
    _load_params(model, opt.golden_params(ocfg, int(z["seed"])))
    flat = FlatParams(model)
    named = dict(model.named_parameters())

    # ---- supervised branch: losses + gradients vs the reference
    perm = opt.SeededPerm(77)
    sampling.set_key_source(perm_key_source(perm))
    try:
        flat.zero_grad()
        losses, _, _, _ = model(_gpu_records(z, "sup", 2), branch="supervised")
        assert perm.log == list(z["sup_perm_log"]), f"label counts differ: {perm.log} vs {list(z['sup_perm_log'])}"
        for k, v in losses.items():
            close(v.detach().cpu(), z["sup_" + k], 1e-4, 1e-6, "sup " + k)
        sum(losses.values()).backward()
        _grad_check(z, "supgrad", named)

        # ---- teacher branch
        sampling.set_key_source(perm_key_source(opt.SeededPerm(78)))
        with torch.no_grad():
            _, prop_rpn, prop_roih, pred = model(_gpu_records(z, "weak", 2), branch="unsup_data_weak")
        for i in range(2):
            ref_b = z[f"t_rpn{i}_proposal_boxes"]
            assert abs(len(prop_rpn[i]) - len(ref_b)) <= 2, f"proposal count {len(prop_rpn[i])} vs {len(ref_b)}"
            zero = np.zeros(len(prop_rpn[i]), np.int64)
            frac, idx = match_detections(prop_rpn[i].proposal_boxes.tensor.cpu(), zero, ref_b, np.zeros(len(ref_b), np.int64))
            assert frac >= 0.97, f"rpn proposals matched {frac:.3f}"
            ok = idx >= 0
            close(prop_rpn[i].objectness_logits.cpu()[idx[ok]], z[f"t_rpn{i}_objectness_logits"][ok], 5e-4, 1e-5, "rpn scores")
            # detections: scores that differ by fp32 noise may swap ranks -> compare order-insensitively
            frac, idx = match_detections(prop_roih[i].pred_boxes.tensor.cpu(), prop_roih[i].pred_classes.cpu(),
                                         z[f"t_roih{i}_pred_boxes"], z[f"t_roih{i}_pred_classes"])
            assert len(prop_roih[i]) == len(z[f"t_roih{i}_scores"]) and frac >= 0.97, f"matched {frac:.3f}"
            ok = idx >= 0
            mine = idx[ok]
            close(prop_roih[i].scores.cpu()[mine], z[f"t_roih{i}_scores"][ok], 5e-4, 1e-6, "det scores")
            close(prop_roih[i].scores_logists.cpu()[mine], z[f"t_roih{i}_scores_logists"][ok], 1e-3, 5e-4, "det logits")
            close(prop_roih[i].boxes_sigma.cpu()[mine], z[f"t_roih{i}_boxes_sigma"][ok], 1e-3, 5e-4, "det sigma")
        assert pred[0].shape == z["t_pred_scores"].shape or abs(pred[0].shape[0] - z["t_pred_scores"].shape[0]) <= 4

        # ---- unsupervised branch fed with the REFERENCE's pseudo labels (the fixture's teacher outputs)
        from probabilisticteacher_amd.structures import Boxes, FreeInstances
        strong = _gpu_records(z, "strong", 2)
        for i, r in enumerate(strong):
            h, w = r["image"].shape[-2:]
            inst = FreeInstances((h, w))
            inst.pseudo_boxes = Boxes(torch.from_numpy(z[f"t_roih{i}_pred_boxes"]))
            inst.scores_logists = torch.from_numpy(z[f"t_roih{i}_scores_logists"])
            inst.boxes_sigma = torch.from_numpy(z[f"t_roih{i}_boxes_sigma"])
            r["instances"] = inst
        flat.zero_grad()
        sampling.set_key_source(perm_key_source(opt.SeededPerm(79)))
        losses_u, _, _, _ = model(strong, branch="unsupervised", danchor=True)
        for k, v in losses_u.items():
            close(v.detach().cpu(), z["unsup_" + k], 1e-4, 1e-6, "unsup " + k)
        sum(losses_u.values()).backward()
        _grad_check(z, "unsupgrad", named)
    finally:
        sampling.set_key_source(None)


def test_run_step_matches_reference_golden():
    """Three real PTrainer.run_step iterations of the reference (burn-in, EMA copy + mutual, EMA + mutual) replayed
    on the HIP trainer: metrics and parameter probes."""
    from probabilisticteacher_amd.engine import PTrainer
    from probabilisticteacher_amd.modeling import sampling
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    z = load("run_step")
    K, tau, B = int(z["K"]), tuple(float(v) for v in z["tau"]), int(z["B"])
    cfg = _cfg(K, "DifferentiableAnchorGenerator", tau, burn=1)
    ocfg = opt.Cfg(num_classes=K, anchor_generator="DifferentiableAnchorGenerator", tau=tau, burn_up_step=1)
    ratios = []

    class ReplayTrainer(PTrainer):
        """Test double: records the teacher's pseudo labels and hands the student the REFERENCE's (from the fixture),
        so that student-side quantities are compared on identical inputs; the teacher's own are compared below."""
        override = None
        mine = None

        def process_pseudo_label(self, proposals, proposal_type, psedo_label_method=""):
            out, n = super().process_pseudo_label(proposals, proposal_type, psedo_label_method)
            self.mine = out
            return (self.override, n) if self.override is not None else (out, n)

    tr = ReplayTrainer(cfg, ratio_fn=lambda: ratios.pop(0))
    _load_params(tr.model, opt.golden_params(ocfg, int(z["seed"])))
    _load_params(tr.model_teacher, opt.golden_params(ocfg, int(z["teacher_seed"])))
    probes = sorted({k.split("_s_sum_")[1] for k in z.files if "_s_sum_" in k})
    try:
        for it in range(3):
            data = tuple(_gpu_records(z, f"it{it}_{nm}", B) for nm in ("lq", "lk", "uq", "uk"))
            ratios[:] = [float(v) for v in z[f"it{it}_ratios"]]
            tr.override = None
            if f"it{it}_pseudo0_pseudo_boxes" in z.files:
                ov = []
                for i in range(B):
                    h, w = data[3][i]["image"].shape[-2:]
                    inst = FreeInstances((h, w))
                    inst.pseudo_boxes = Boxes(torch.from_numpy(z[f"it{it}_pseudo{i}_pseudo_boxes"]).to(DEV))
                    inst.scores_logists = torch.from_numpy(z[f"it{it}_pseudo{i}_scores_logists"]).to(DEV)
                    inst.boxes_sigma = torch.from_numpy(z[f"it{it}_pseudo{i}_boxes_sigma"]).to(DEV)
                    ov.append(inst)
                tr.override = ov
            sampling.set_key_source(perm_key_source(opt.SeededPerm(500 + it)))
            m = tr.run_step(data)
            if tr.override is not None:
                for mine, ref in zip(tr.mine, tr.override):
                    assert len(mine) == len(ref)
                    ca = mine.scores_logists[:, :-1].argmax(1).cpu() * 0      # class-agnostic match on boxes
                    frac, idx = match_detections(mine.pseudo_boxes.tensor.cpu(), ca, ref.pseudo_boxes.tensor.cpu(),
                                                 ca, box_tol=5e-2)
                    assert frac >= 0.95, f"pseudo boxes matched {frac:.3f}"
            for k in z.files:
                if k.startswith(f"it{it}_m_"):
                    # it 0 runs on identical weights (1e-4); later iterations inherit the fp32 summation-order
                    # differences (MFMA k-order vs oneDNN blocking) through the optimiser step: 1e-3
                    close(torch.tensor(m[k[len(f"it{it}_m_"):]]), z[k], 1e-4 if it == 0 else 1e-3, 1e-6, k)
            ssd, tsd = tr.model.state_dict(), tr.model_teacher.state_dict()
            # parameter sums are cancellation-heavy (25.7 M fc1 weights summing to ~4): identical weights at it 0
            # agree to ~4e-6; afterwards the fp32 summation-order noise of the GEMM / conv kernels compounds
            # through the optimiser (measured 4e-5 .. 3e-4 at it 1-2 with either GEMM kernel)
            sum_atol = 2e-4 if it == 0 else 1e-3
            for k in probes:
                close(ssd[k].double().sum().cpu(), z[f"it{it}_s_sum_{k}"], 1e-5, sum_atol, f"student sum {k}")
                close(ssd[k].flatten()[:16].cpu(), z[f"it{it}_s_head_{k}"], 1e-4, 1e-6, f"student head {k}")
                close(tsd[k].double().sum().cpu(), z[f"it{it}_t_sum_{k}"], 1e-5, sum_atol, f"teacher sum {k}")
                close(tsd[k].flatten()[:16].cpu(), z[f"it{it}_t_head_{k}"], 1e-4, 1e-6, f"teacher head {k}")
    finally:
        sampling.set_key_source(None)


def test_run_step_long_reference_golden_replayed_on_hip(capsys):
    """Round 6: the FOURTEEN real reference iterations of tests/golden/run_step_long.npz (3 burn-in,
    the keep_rate = 0 copy, 10 EMA-0.9996 mutual-learning steps, the reference's LR warm-up; tools/gen_golden.py::
    gen_run_step_long) on the HIP trainer, EVERY iteration against the reference's own numbers.
    Two fp32 implementations that start from identical weights drift apart exponentially under SGD (measured here: worst metric
    deviation 1e-7, 9e-6, 8e-5, 4e-4, 1e-3 over iterations 0 .. 4 -- each side samples its own proposals), so a free-running
    14-step replay cannot be held to a tolerance.  Instead the ORACLE runs alongside (tests/test_oracle_golden.py holds it to the
    reference at 2e-4 / 1e-5 over all 14 iterations) and carries the state: before iteration `it` the HIP trainer is given the
    oracle's student, teacher, momentum buffers and iteration counter, takes the step -- EMA update, teacher pass, pseudo labels
    (the student consumes the REFERENCE's, from the fixture), both student branches, backward, clip + SGD under the warm-up LR --
    and its metrics and 7 + 7 parameter probes are compared with the REFERENCE's at the single-step bar: metrics 1e-3 (each
    side's own proposal sample from weights that agree to ~1e-6; iteration 0, bit-identical weights: 1e-4), probe heads 1e-4."""
    from probabilisticteacher_amd.engine import PTrainer
    from probabilisticteacher_amd.modeling import sampling
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    from probabilisticteacher_amd.solver import lr_at
    z = load("run_step_long")
    K, tau, B, burn, iters = int(z["K"]), tuple(float(v) for v in z["tau"]), int(z["B"]), int(z["burn"]), int(z["iters"])
    cfg = _cfg(K, "DifferentiableAnchorGenerator", tau, burn=burn)
    assert cfg.SOLVER.BASE_LR == float(z["base_lr"]) and cfg.SOLVER.WARMUP_ITERS == int(z["warmup_iters"])
    assert cfg.UNSUPNET.EMA_KEEP_RATE == float(z["ema_keep_rate"])
    ocfg = opt.Cfg(num_classes=K, anchor_generator="DifferentiableAnchorGenerator", tau=tau, burn_up_step=burn)
    ratios = []

    class ReplayTrainer(PTrainer):
        override = None
        mine = None

        def process_pseudo_label(self, proposals, proposal_type, psedo_label_method=""):
            out, n = super().process_pseudo_label(proposals, proposal_type, psedo_label_method)
            self.mine = out
            return (self.override, n) if self.override is not None else (out, n)

    tr = ReplayTrainer(cfg, ratio_fn=lambda: ratios.pop(0))
    state = {"student": opt.golden_params(ocfg, int(z["seed"])), "teacher": opt.golden_params(ocfg, int(z["teacher_seed"])),
             "bufs": {}, "iter": 0}
    probes = sorted({k.split("_s_sum_")[1] for k in z.files if "_s_sum_" in k})
    trainable = [n for n, p in tr.student.params.items() if p.requires_grad]

    def inject(st):
        """the oracle's state -> the HIP trainer (parameters through the state dicts, momentum into the flat buffer)"""
        _load_params(tr.model, st["student"])
        _load_params(tr.model_teacher, st["teacher"])
        tr.iter = st["iter"]
        tr._first_step = not st["bufs"]
        if st["bufs"]:
            assert set(st["bufs"]) == set(trainable), set(st["bufs"]) ^ set(trainable)
            with torch.no_grad():
                for n in trainable:
                    off, k = tr.student.index[n]
                    tr.momentum_buf[off:off + k].copy_(st["bufs"][n].reshape(-1))

    report = []
    # the oracle's arithmetic as in the dev container that chose the fixture's data seed (8 oneDNN threads: another thread count is
    # another summation order, and over 14 iterations one flipped index decision moves a loss by ~1 % -- tools/gen_golden.py)
    torch.set_num_threads(8)
    compared = 0
    try:
        for it in range(iters):
            inject(state)
            data = tuple(_gpu_records(z, f"it{it}_{nm}", B) for nm in ("lq", "lk", "uq", "uk"))
            odata = tuple(records(z, f"it{it}_{nm}", B) for nm in ("lq", "lk", "uq", "uk"))
            rr = [float(v) for v in z[f"it{it}_ratios"]]
            ratios[:] = rr
            tr.override, o_override = None, None
            if f"it{it}_pseudo0_pseudo_boxes" in z.files:
                ov, oov = [], []
                for i in range(B):
                    h, w = data[3][i]["image"].shape[-2:]
                    inst, oinst = FreeInstances((h, w)), opt.FreeInstances((h, w))
                    pb, sl, bs = (torch.from_numpy(z[f"it{it}_pseudo{i}_{f}"]) for f in ("pseudo_boxes", "scores_logists", "boxes_sigma"))
                    inst.pseudo_boxes, inst.scores_logists, inst.boxes_sigma = Boxes(pb.to(DEV)), sl.to(DEV), bs.to(DEV)
                    oinst.pseudo_boxes, oinst.scores_logists, oinst.boxes_sigma = d2.Boxes(pb.clone()), sl.clone(), bs.clone()
                    ov.append(inst)
                    oov.append(oinst)
                tr.override, o_override = ov, oov
            sampling.set_key_source(perm_key_source(opt.SeededPerm(700 + it)))
            assert abs(lr_at(cfg, it) - float(z[f"it{it}_lr"])) <= 1e-12
            m = tr.run_step(data)
            # the oracle takes the same step from the same state: the state carrier of the next iteration
            om = opt.run_step(ocfg, state, odata, {"label": rr, "unlabel": []} if it < burn else {"unlabel": rr[:B], "label": rr[B:2 * B]},
                              perm_fn=opt.SeededPerm(700 + it), pseudo_override=o_override)
            frac = None
            if tr.override is not None:
                for mine, ref in zip(tr.mine, tr.override):
                    frac, _ = match_detections(mine.pseudo_boxes.tensor.cpu(), np.zeros(len(mine), np.int64),
                                               ref.pseudo_boxes.tensor.cpu(), np.zeros(len(ref), np.int64), box_tol=5e-2)
                    assert frac >= 0.95, f"iteration {it}: the HIP teacher's pseudo boxes matched {frac:.3f} of the reference teacher's"
            # the state carrier must still BE the reference: on a host whose oneDNN sums in another order the oracle may leave the
            # reference at some iteration through one flipped index decision (first GPU-box run: iteration 9, one term off by 3e-3,
            # all others at 1e-5); from there on its state is no longer the reference's and the comparison stops -- after at least
            # 8 iterations (the copy step and >= 4 EMA steps among them)
            carrier_off = [k for k in z.files if k.startswith(f"it{it}_m_") and not np.isnan(z[k]) and
                           abs(om[k[len(f"it{it}_m_"):]] - float(z[k])) > 2e-4 * abs(float(z[k])) + 1e-6]
            if carrier_off:
                report.append(f"it {it}: the oracle left the reference on this host ({carrier_off[0]}): stopped")
                break
            worst = 0.0
            for k in z.files:
                if k.startswith(f"it{it}_m_"):
                    name = k[len(f"it{it}_m_"):]
                    if np.isnan(z[k]):
                        assert math.isnan(m[name]) and math.isnan(om[name]), f"{k}: reference NaN (empty mean), HIP {m[name]}, oracle {om[name]}"
                        continue
                    worst = max(worst, abs(m[name] - float(z[k])) / (abs(float(z[k])) + 1e-6))
                    close(torch.tensor(m[name]), z[k], 1e-4 if it == 0 else 1e-3, 1e-6, k)
            compared += 1
            ssd, tsd = tr.model.state_dict(), tr.model_teacher.state_dict()
            for k in probes:
                close(ssd[k].double().sum().cpu(), z[f"it{it}_s_sum_{k}"], 1e-5, 1e-3, f"it {it} student sum {k}")
                close(ssd[k].flatten()[:16].cpu(), z[f"it{it}_s_head_{k}"], 1e-4, 1e-6, f"it {it} student head {k}")
                close(tsd[k].double().sum().cpu(), z[f"it{it}_t_sum_{k}"], 1e-5, 1e-3, f"it {it} teacher sum {k}")
                close(tsd[k].flatten()[:16].cpu(), z[f"it{it}_t_head_{k}"], 1e-4, 1e-6, f"it {it} teacher head {k}")
            report.append(f"it {it}: worst metric deviation {worst:.2e}" + (f", pseudo boxes matched {frac:.3f}" if frac is not None else ""))
    finally:
        sampling.set_key_source(None)
        with capsys.disabled():
            print("\n[run_step_long on HIP, state carried by the oracle] " + "; ".join(report))
    assert compared >= 8, f"only {compared} iterations compared before the state carrier left the reference"


def test_full_size_1333x800_backbone_and_rpn_vs_oracle():
    """BASELINE-size parity: one synthetic 1333x800 image through the HIP VGG16 + RPN head vs the CPU oracle
    (features rtol 2e-4 of the max, anchors bit-exact, objectness / deltas 1e-3), plus size-independent properties
    of the proposal stage (scores sorted, boxes inside the image, NMS idempotent)."""
    from probabilisticteacher_amd import modeling, ops
    g = torch.Generator().manual_seed(9)
    K = 8
    cfg = _cfg(K, "DefaultAnchorGenerator", (0.25, 0.25))
    ocfg = opt.Cfg(num_classes=K)
    params = opt.golden_params(ocfg, 2)
    model = modeling.build_model(cfg).train()
    _load_params(model, params)
    img = torch.randint(0, 256, (3, 800, 1333), generator=g, dtype=torch.uint8)
    with torch.no_grad():
        images = model.preprocess_image([{"image": img}])
        feat = model.backbone(images.tensor)["vgg_block5"]
        torch.set_num_threads(min(32, torch.get_num_threads()))
        oimg = opt.preprocess_image(ocfg, [{"image": img}])
        assert torch.equal(images.tensor.cpu(), oimg.tensor)
        ofeat = opt.vgg_forward(params, oimg.tensor)
        assert feat.shape == (1, 512, 50, 83)
        scale = float(ofeat.abs().max())
        assert float((feat.cpu() - ofeat).abs().max()) <= 2e-4 * scale, "block5 features"
        obj, deltas = model.proposal_generator.rpn_head([feat])
        oobj, odl = opt.rpn_head_forward(params, ofeat)
        logits = obj[0].permute(0, 2, 3, 1).reshape(1, -1)
        close(logits.cpu(), oobj, 1e-3, 1e-3 * float(oobj.abs().max()), "objectness")
        d8 = deltas[0].view(1, 9, 8, 50, 83).permute(0, 3, 4, 1, 2).reshape(1, -1, 8)
        close(d8.cpu(), odl, 1e-3, 1e-3 * float(odl.abs().max()), "anchor deltas")
        anchors = model.proposal_generator.anchor_generator([feat])[0].tensor
        assert torch.equal(anchors.cpu(), opt.make_anchors(ocfg, params, (50, 83), False)) and len(anchors) == 37350
        props, _ = model.proposal_generator(images, {"vgg_block5": feat}, None, compute_loss=False)
        p = props[0]
        b, s = p.proposal_boxes.tensor, p.objectness_logits
        assert 0 < len(p) <= 2000
        assert bool((s[:-1] >= s[1:]).all()), "proposals are sorted by (rescored) score"
        assert bool((b[:, 0] >= 0).all() and (b[:, 1] >= 0).all() and (b[:, 2] <= 1333).all() and (b[:, 3] <= 800).all())
        assert bool(((b[:, 2] - b[:, 0]) > 0).all() and ((b[:, 3] - b[:, 1]) > 0).all())
        seg = torch.tensor([0, len(p)], dtype=torch.int32, device=DEV)
        keep, cnt = ops.nms_batched(b.contiguous(), seg, len(p), 0.7, len(p))
        assert int(cnt[0]) == len(p) and torch.equal(keep[0].cpu().long(), torch.arange(len(p))), "NMS is idempotent"


def test_edge_cases_empty_gt_and_no_pseudo_matches():
    """Edge cases of the reference's control flow: an image without ground truth in the supervised branch
    (rpn.py:435-437, roi_heads.py:238-242) and an unsupervised image whose pseudo boxes match no proposal."""
    from probabilisticteacher_amd import modeling
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    K = 8
    cfg = _cfg(K, "DefaultAnchorGenerator", (0.25, 0.25))
    ocfg = opt.Cfg(num_classes=K)
    params = opt.golden_params(ocfg, 5)
    model = modeling.build_model(cfg).train()
    _load_params(model, params)
    g = torch.Generator().manual_seed(1)
    recs, orecs = [], []
    for i in range(2):
        img = torch.randint(0, 256, (3, 96, 128), generator=g, dtype=torch.uint8)
        boxes = torch.tensor([[10.0, 12.0, 70.0, 80.0]]) if i == 0 else torch.zeros((0, 4))
        cls = torch.tensor([3]) if i == 0 else torch.zeros(0, dtype=torch.int64)
        a, b = FreeInstances((96, 128)), opt.FreeInstances((96, 128))
        a.gt_boxes, a.gt_classes = Boxes(boxes.clone()), cls.clone()
        b.gt_boxes, b.gt_classes = d2.Boxes(boxes.clone()), cls.clone()
        recs.append({"image": img, "instances": a})
        orecs.append({"image": img, "instances": b})
    from probabilisticteacher_amd.modeling import sampling
    sampling.set_key_source(perm_key_source(opt.SeededPerm(11)))
    try:
        got, _, _, _ = model(recs, branch="supervised")
    finally:
        sampling.set_key_source(None)
    ref, _, _, _ = opt.model_forward(ocfg, params, orecs, "supervised", perm_fn=opt.SeededPerm(11))
    for k in ref:
        close(got[k].detach().cpu(), ref[k].detach(), 2e-4, 1e-6, "empty-gt " + k)
    sum(got.values()).backward()
    # unsupervised with a pseudo box far away from every proposal -> no ROI survives: cls loss is NaN (0/0) exactly
    # like the reference (fast_rcnn.py:209), the box loss too (mean of an empty tensor)
    inst = FreeInstances((96, 128))
    inst.pseudo_boxes = Boxes(torch.tensor([[0.0, 0.0, 1.0, 1.0]]))
    inst.scores_logists = torch.zeros(1, K + 1)
    inst.boxes_sigma = torch.zeros(1, 4)
    un = [{"image": recs[0]["image"], "instances": inst}]
    lu, _, _, _ = model(un, branch="unsupervised", danchor=True)
    assert torch.isnan(lu["loss_cls"]) and torch.isfinite(lu["loss_rpn_cls"])


def test_eval_mode_inference_matches_oracle():
    """`model.eval(); model(batched_inputs)` (rcnn.py:33-34 -> D2 inference + detector_postprocess):
    test-time RPN top-k, ROI inference and the rescaling of the boxes to the record's height / width, against the
    CPU oracle (order-insensitive: near-equal scores may swap ranks)."""
    from probabilisticteacher_amd import modeling
    K = 8
    cfg = _cfg(K, "DefaultAnchorGenerator", (0.25, 0.25))
    ocfg = opt.Cfg(num_classes=K)
    params = opt.golden_params(ocfg, 9)
    model = modeling.build_model(cfg).eval()
    _load_params(model, params)
    g = torch.Generator().manual_seed(3)
    recs = []
    for i, (h, w) in enumerate([(96, 128), (80, 112)]):
        img = torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8)
        recs.append({"image": img, "height": 2 * h + i, "width": 3 * w})          # output size != input size
    with torch.no_grad():
        got = model(recs)
    ref = opt.model_inference(ocfg, params, recs)
    assert len(got) == len(ref) == 2
    for a, b, rec in zip(got, ref, recs):
        a, b = a["instances"], b["instances"]
        assert a.image_size == b.image_size == (rec["height"], rec["width"])
        assert abs(len(a) - len(b)) <= max(2, len(b) // 20)
        if len(b):
            frac, idx = match_detections(a.pred_boxes.tensor.cpu(), a.pred_classes.cpu(), b.pred_boxes.tensor,
                                         b.pred_classes, box_tol=5e-2)
            assert frac >= 0.95, f"eval detections matched {frac:.3f}"
            bt = a.pred_boxes.tensor.cpu()
            assert (bt[:, 0::2] <= rec["width"]).all() and (bt[:, 1::2] <= rec["height"]).all() and (bt >= 0).all()


def test_supervised_and_unsup_branches_with_keyed_sampling_match_oracle():
    """The production sampler (random keys + top-k, no per-image host sync) against the oracle: both sides consume the
    same key stream (oracle KeyedPerm), so the sampled anchors / proposals -- and with them all four supervised losses
    -- must agree; the unsupervised branch exercises the batched (single-nonzero) positive gathers."""
    from probabilisticteacher_amd import modeling
    from probabilisticteacher_amd.modeling import sampling
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    K = 8
    cfg = _cfg(K, "DefaultAnchorGenerator", (0.25, 0.25))
    ocfg = opt.Cfg(num_classes=K)
    params = opt.golden_params(ocfg, 21)
    model = modeling.build_model(cfg).train()
    _load_params(model, params)
    g = torch.Generator().manual_seed(4)
    recs, orecs = [], []
    for i in range(3):
        img = torch.randint(0, 256, (3, 128, 160), generator=g, dtype=torch.uint8)
        m = [4, 1, 0][i]                                                  # one image without ground truth
        xy = torch.rand(m, 2, generator=g) * torch.tensor([90.0, 70.0])
        wh = 20 + torch.rand(m, 2, generator=g) * 50
        boxes = torch.cat([xy, xy + wh], 1)
        cls = torch.randint(0, K, (m,), generator=g)
        a, b = FreeInstances((128, 160)), opt.FreeInstances((128, 160))
        a.gt_boxes, a.gt_classes = Boxes(boxes.clone()), cls.clone()
        b.gt_boxes, b.gt_classes = d2.Boxes(boxes.clone()), cls.clone()
        recs.append({"image": img, "instances": a})
        orecs.append({"image": img, "instances": b})
    kp = opt.KeyedPerm(31)
    sampling.set_key_source(keyed_perm_source(kp))
    try:
        got, _, _, _ = model(recs, branch="supervised")
    finally:
        sampling.set_key_source(None)
    kp.start_replay()
    ref, _, _, _ = opt.model_forward(ocfg, params, orecs, "supervised", perm_fn=kp)
    assert not kp.replay, "the oracle must consume every key row the product drew"
    for k in ref:
        close(got[k].detach().cpu(), ref[k].detach(), 2e-4, 1e-6, "keyed sampling " + k)
    sum(got.values()).backward()
    # unsupervised branch (no randomness): pseudo labels on two images, one of them with a single far-away box
    un, oun = [], []
    for i in range(2):
        pb = torch.tensor([[20.0, 30.0, 90.0, 100.0], [60.0, 10.0, 150.0, 90.0]]) if i == 0 else torch.tensor([[5.0, 5.0, 60.0, 70.0]])
        lg = torch.randn(len(pb), K + 1, generator=g)
        sg = torch.randn(len(pb), 4, generator=g)
        a, b = FreeInstances((128, 160)), opt.FreeInstances((128, 160))
        a.pseudo_boxes, a.scores_logists, a.boxes_sigma = Boxes(pb.clone()), lg.clone(), sg.clone()
        b.pseudo_boxes, b.scores_logists, b.boxes_sigma = d2.Boxes(pb.clone()), lg.clone(), sg.clone()
        un.append({"image": recs[i]["image"], "instances": a})
        oun.append({"image": recs[i]["image"], "instances": b})
    gu, _, _, _ = model(un, branch="unsupervised", danchor=True)
    ru, _, _, _ = opt.model_forward(ocfg, params, oun, "unsupervised", danchor=True)
    for k in ru:
        if torch.isnan(ru[k]):
            assert torch.isnan(gu[k]), k
        else:
            close(gu[k].detach().cpu(), ru[k].detach(), 5e-4, 1e-6, "unsup " + k)


def test_joint_student_pass_equals_separate_passes():
    """forward_joint (one backbone / RPN-head pass over supervised + unsupervised images) against the two separate
    `model(...)` calls of trainer.py:341,353-355 on the same inputs and the same sampler keys: identical losses and
    matching gradients (the weight gradients only differ in summation order)."""
    from probabilisticteacher_amd import modeling
    from probabilisticteacher_amd.engine.flat import FlatParams
    from probabilisticteacher_amd.modeling import sampling
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    K = 8
    cfg = _cfg(K, "DifferentiableAnchorGenerator", (0.5, 0.5))
    ocfg = opt.Cfg(num_classes=K, anchor_generator="DifferentiableAnchorGenerator")
    params = opt.golden_params(ocfg, 13)
    model = modeling.build_model(cfg).train()
    _load_params(model, params)
    flat = FlatParams(model)
    g = torch.Generator().manual_seed(8)
    sup, un = [], []
    for i in range(3):
        img = torch.randint(0, 256, (3, 112, 144), generator=g, dtype=torch.uint8)
        m = 1 + i
        xy = torch.rand(m, 2, generator=g) * torch.tensor([80.0, 60.0])
        a = FreeInstances((112, 144))
        a.gt_boxes, a.gt_classes = Boxes(torch.cat([xy, xy + 25 + torch.rand(m, 2, generator=g) * 30], 1)), torch.randint(0, K, (m,), generator=g)
        sup.append({"image": img, "instances": a})
    for i in range(2):
        img = torch.randint(0, 256, (3, 112, 144), generator=g, dtype=torch.uint8)
        a = FreeInstances((112, 144))
        a.pseudo_boxes = Boxes(torch.tensor([[20.0, 30.0, 90.0, 100.0], [60.0, 10.0, 140.0, 90.0]])[: 2 - i])
        a.scores_logists = torch.randn(2 - i, K + 1, generator=g)
        a.boxes_sigma = torch.randn(2 - i, 4, generator=g)
        un.append({"image": img, "instances": a})
    assert model.can_run_jointly(sup, un)

    def run(joint):
        kp = opt.KeyedPerm(5)
        sampling.set_key_source(keyed_perm_source(kp))
        flat.zero_grad()
        try:
            if joint:
                ls, lu = model.forward_joint(sup, un, danchor=True)
            else:
                ls, _, _, _ = model(sup, branch="supervised")
                lu, _, _, _ = model(un, branch="unsupervised", danchor=True)
        finally:
            sampling.set_key_source(None)
        (sum(ls.values()) + sum(lu.values())).backward()
        return ({k: float(v) for k, v in ls.items()}, {k: float(v) for k, v in lu.items()}, flat.grad.clone())
    s1, u1, g1 = run(False)
    s2, u2, g2 = run(True)
    for a, b in ((s1, s2), (u1, u2)):
        assert a.keys() == b.keys()
        for k in a:
            close(b[k], a[k], 1e-6, 1e-7, "joint vs separate " + k)
    close(g2.cpu(), g1.cpu(), 1e-4, 1e-5 * float(g1.abs().max()), "joint vs separate gradients")
    for rec in un:
        rec["image"] = torch.randint(0, 256, (3, 96, 144), generator=g, dtype=torch.uint8)
    assert not model.can_run_jointly(sup, un), "different canvases must fall back to separate passes"


def test_baseline_config0_800x600_supervised_step_vs_oracle():
    """BASELINE.json configs[0] (Guassian-RCNN-VGG.yaml, 2 synthetic 800x600 images, one supervised iteration) on the HIP
    trainer against the CPU oracle's run_step with the same sampler keys and shrink ratios: the four losses to 1e-4
    (north_star: "loss parity vs CPU reference to 1e-4"), the gradient norm, and the updated parameters."""
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer
    from probabilisticteacher_amd.modeling import sampling
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    cfg = setup_cfg("configs/Guassian-RCNN-VGG.yaml", ["MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", "",
                                                      "SOLVER.IMG_PER_BATCH_LABEL", 1, "SOLVER.IMG_PER_BATCH_UNLABEL", 1])
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    ocfg = opt.Cfg(num_classes=K, burn_up_step=int(cfg.UNSUPNET.BURN_UP_STEP))
    assert cfg.MODEL.ANCHOR_GENERATOR.NAME == ocfg.anchor_generator and ocfg.burn_up_step > 0
    params = opt.golden_params(ocfg, 17)
    ratios_q = [0.8, 0.65]
    it = iter(list(ratios_q))
    tr = PTrainer(cfg, ratio_fn=lambda: next(it))
    _load_params(tr.model, params)
    _load_params(tr.model_teacher, params)
    g = torch.Generator().manual_seed(12)
    recs, orecs = [], []
    for i in range(2):                                  # label_q[0] and label_k[0]: the burn-in batch of 2 images
        img = torch.randint(0, 256, (3, 600, 800), generator=g, dtype=torch.uint8)
        m = 3 + i
        xy = torch.rand(m, 2, generator=g) * torch.tensor([500.0, 350.0])
        boxes = torch.cat([xy, xy + 40 + torch.rand(m, 2, generator=g) * 200], 1)
        cls = torch.randint(0, K, (m,), generator=g)
        a, b = FreeInstances((600, 800)), opt.FreeInstances((600, 800))
        a.gt_boxes, a.gt_classes = Boxes(boxes.clone()), cls.clone()
        b.gt_boxes, b.gt_classes = d2.Boxes(boxes.clone()), cls.clone()
        recs.append({"image": img, "height": 600, "width": 800, "instances": a})
        orecs.append({"image": img, "height": 600, "width": 800, "instances": b})
    kp = opt.KeyedPerm(41, strict=False)
    sampling.set_key_source(keyed_perm_source(kp))
    try:
        m = tr.run_step(([recs[0]], [recs[1]], [recs[0]], [recs[1]]))
    finally:
        sampling.set_key_source(None)
    kp.start_replay()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    state = {"student": {k: v.clone() for k, v in params.items()}, "teacher": {k: v.clone() for k, v in params.items()},
             "bufs": {}, "iter": 0}
    om = opt.run_step(ocfg, state, ([orecs[0]], [orecs[1]], [orecs[0]], [orecs[1]]), {"label": ratios_q, "unlabel": []},
                      perm_fn=kp)
    # the anchor sample (fixed 16 650 candidates per image) is always identical; the ROI sample is identical unless the
    # two sides kept a different number of proposals (one box more or less surviving NMS at this scale), in which case
    # the ROI losses are only statistically equal
    for k in ("loss_rpn_cls", "loss_rpn_loc"):
        close(torch.tensor(m[k]), torch.tensor(om[k]), 1e-4, 1e-6, "configs[0] " + k)
    roi_tol = 5e-2 if kp.mismatch else 1e-4
    for k in ("loss_cls", "loss_box_reg"):
        close(torch.tensor(m[k]), torch.tensor(om[k]), roi_tol, 1e-6, "configs[0] " + k)
    if kp.mismatch:
        return
    close(torch.tensor(m["total_loss"]), torch.tensor(om["total_loss"]), 1e-4, 1e-6, "configs[0] total_loss")
    sd = tr.model.state_dict()
    for k in ("roi_heads.box_predictor.cls_score.weight", "proposal_generator.rpn_head.conv.bias",
              "backbone.vgg_block5.0.conv3.weight", "backbone.vgg_block3.0.conv1.weight", "roi_heads.box_head.fc1.bias"):
        ref = state["student"][k].detach()
        assert not torch.equal(ref, params[k]), k + " must have been updated"
        close(sd[k].cpu(), ref, 1e-4, 1e-5 * float(ref.abs().max()) + 1e-7, "configs[0] updated " + k)


def test_voc_evaluation_of_the_eval_path_matches_the_oracle_detections():
    """SURVEY.md 8f-2 end to end: `PTrainer.test` = eval-mode inference (rcnn.py:33-34) + the VOC evaluator
    (trainer.py:127-137) on a few labelled synthetic records, against the same evaluator fed with the CPU oracle's
    detections, and -- the evaluator itself -- against the INDEPENDENT protocol restatement oracle/voc.py on the HIP
    detections: AP / AP50 / AP75 to 1e-9.  (HIP vs oracle MODEL under the metric: score ranks that differ in the last ulp can
    move a detection across a precision step, 2 points; the model outputs themselves are compared in
    test_eval_mode_inference_matches_oracle.)"""
    from probabilisticteacher_amd import modeling
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer
    from probabilisticteacher_amd.evaluation import PascalVOCDetectionEvaluator
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    K = 8
    cfg = setup_cfg("configs/pt/final_c2f.yaml", ["MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", "", "TEST.EVALUATOR", "VOCeval"])
    ocfg = opt.Cfg(num_classes=K)
    params = opt.golden_params(ocfg, 9)
    model = modeling.build_model(cfg).train()
    _load_params(model, params)
    g = torch.Generator().manual_seed(3)
    batches = []
    for bi in range(2):
        recs = []
        for i in range(2):
            h, w = 96 + 16 * i, 128
            img = torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8)
            xy = torch.rand(3, 2, generator=g) * torch.tensor([60.0, 40.0])
            gt = FreeInstances((h, w))
            gt.gt_boxes, gt.gt_classes = Boxes(torch.cat([xy, xy + 20 + torch.rand(3, 2, generator=g) * 30], 1)), torch.randint(0, K, (3,), generator=g)
            recs.append({"image": img, "height": h, "width": w, "image_id": 10 * bi + i, "instances": gt})
        batches.append(recs)
    names = [f"c{i}" for i in range(K)]
    res = PTrainer.test(cfg, model, batches, names)
    assert model.training, "the previous mode is restored"
    # the product evaluator vs the independent restatement, on the HIP model's own detections
    from oracle import voc
    model.eval()
    dets, gts = [], {}
    with torch.no_grad():
        for recs in batches:
            for r, o in zip(recs, model(recs)):
                inst, gt = o["instances"], r["instances"]
                dets += [(r["image_id"], int(c), float(s), *[float(v) for v in b]) for b, c, s in
                         zip(inst.pred_boxes.tensor.cpu().numpy(), inst.pred_classes.cpu().tolist(), inst.scores.cpu().tolist())]
                gts[r["image_id"]] = [(int(c), *[float(v) for v in b], False) for b, c in
                                      zip(gt.gt_boxes.tensor.cpu().numpy(), gt.gt_classes.cpu().tolist())]
    model.train()
    ind = voc.evaluate(dets, gts, K)
    for k in ("AP", "AP50", "AP75"):
        assert abs(res["bbox"][k] - ind[k]) < 1e-9, (k, res["bbox"], ind)
    ev = PascalVOCDetectionEvaluator(names)
    for recs in batches:
        ev.process(recs, opt.model_inference(ocfg, params, [{k: v for k, v in r.items() if k != "instances"} for r in recs]))
    ref = ev.evaluate()
    for k in ("AP", "AP50", "AP75"):
        assert np.isfinite(res["bbox"][k]) and abs(res["bbox"][k] - ref["bbox"][k]) <= 2.0, (k, res["bbox"], ref["bbox"])
    with pytest.raises(ValueError):
        PTrainer.build_evaluator(setup_cfg("configs/pt/final_c2f.yaml", ["TEST.EVALUATOR", "nope"]), names)


def test_amp_step_native_bf16_kernels_match_emulated_rounding():
    """SOLVER.AMP.ENABLED (BASELINE configs[4]): one supervised step of the real trainer on the run_step fixture's records,
    (a) on the native bf16-input kernels and (b) with operands rounded by tensor passes on the fp32 kernels, from the same
    weights and sampler keys.  The two differ only in fp32 summation order: RPN terms (anchors are fixed) to 1e-3, ROI terms
    (they inherit proposal-order decisions) to 3e-2, and every updated-parameter probe to the same precision as two fp32
    implementations; and AMP really changes the numbers relative to fp32."""
    from probabilisticteacher_amd import ops
    from probabilisticteacher_amd.engine import PTrainer
    from probabilisticteacher_amd.modeling import sampling
    z = load("run_step")
    K, tau, B = int(z["K"]), tuple(float(v) for v in z["tau"]), int(z["B"])
    ocfg = opt.Cfg(num_classes=K, anchor_generator="DifferentiableAnchorGenerator", tau=tau, burn_up_step=1)
    params = opt.golden_params(ocfg, int(z["seed"]))
    out = {}
    try:
        for mode in ("bf16", "bf16_emulate", None):
            cfg = _cfg(K, "DifferentiableAnchorGenerator", tau, burn=1)
            cfg.defrost() if hasattr(cfg, "defrost") else None
            cfg.SOLVER.AMP.ENABLED = mode is not None
            ratios = [float(v) for v in z["it0_ratios"]]
            tr = PTrainer(cfg, ratio_fn=lambda: ratios.pop(0))
            assert tr.operand_rounding == ("bf16" if mode else None)          # what the config flag selects
            tr.operand_rounding = mode
            _load_params(tr.model, params)
            _load_params(tr.model_teacher, params)
            data = tuple(_gpu_records(z, f"it0_{nm}", B) for nm in ("lq", "lk", "uq", "uk"))
            sampling.set_key_source(perm_key_source(opt.SeededPerm(500)))
            m = tr.run_step(data)
            sd = tr.model.state_dict()
            out[mode] = (m, {k: sd[k].flatten()[:64].cpu().clone() for k in
                             ("backbone.vgg_block3.0.conv1.weight", "proposal_generator.rpn_head.conv.weight",
                              "roi_heads.box_head.fc1.weight", "roi_heads.box_predictor.cls_score.weight")})
    finally:
        sampling.set_key_source(None)
        ops.set_operand_rounding(None)
    (mn, pn), (me, pe), (mf, pf) = out["bf16"], out["bf16_emulate"], out[None]
    for k in ("loss_rpn_cls", "loss_rpn_loc"):
        close(torch.tensor(mn[k]), torch.tensor(me[k]), 1e-3, 1e-5, "native vs emulate " + k)
    for k in ("loss_cls", "loss_box_reg", "grad_norm"):
        close(torch.tensor(mn[k]), torch.tensor(me[k]), 3e-2, 1e-4, "native vs emulate " + k)
    for k in pn:
        close(pn[k], pe[k], 1e-3, 2e-5, "updated parameter " + k)
    assert abs(mn["loss_rpn_cls"] - mf["loss_rpn_cls"]) > 1e-5 or abs(mn["loss_rpn_loc"] - mf["loss_rpn_loc"]) > 1e-5
    for k in ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg"):
        close(torch.tensor(mn[k]), torch.tensor(mf[k]), 5e-2, 1e-3, "bf16 vs fp32 " + k)


def test_rpn_loss_weight_quirk_matches_the_reference():
    """MODEL.RPN.LOSS_WEIGHT = 2, BBOX_REG_LOSS_WEIGHT = 0.5: the product reproduces the reference's double application of the
    RPN loss-weight dict to the supervised losses (rpn.py:141, :254) and none to the unsupervised ones -- against the losses
    of the REAL reference model (tests/golden/rpn_loss_weight.npz)."""
    from probabilisticteacher_amd import modeling
    from probabilisticteacher_amd.modeling import sampling
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    z = load("rpn_loss_weight")
    K, tau = int(z["K"]), tuple(float(v) for v in z["tau"])
    cfg = _cfg(K, "DefaultAnchorGenerator", tau)
    cfg.defrost()
    cfg.MODEL.RPN.LOSS_WEIGHT = float(z["loss_weight"])
    cfg.MODEL.RPN.BBOX_REG_LOSS_WEIGHT = float(z["bbox_reg_loss_weight"])
    cfg.freeze()
    model = modeling.build_model(cfg)
    model.train()
    _load_params(model, opt.golden_params(opt.Cfg(num_classes=K, tau=tau), int(z["seed"])))
    try:
        sampling.set_key_source(perm_key_source(opt.SeededPerm(91)))
        losses, _, _, _ = model(_gpu_records(z, "sup", 2), branch="supervised")
        for k, v in losses.items():
            close(v.detach().cpu(), z["sup_" + k], 1e-4, 1e-6, "sup " + k)
        strong = _gpu_records(z, "strong", 2)
        for i, r in enumerate(strong):
            inst = FreeInstances(tuple(r["image"].shape[-2:]))
            inst.pseudo_boxes = Boxes(torch.from_numpy(z[f"pseudo{i}_pseudo_boxes"]))
            inst.scores_logists = torch.from_numpy(z[f"pseudo{i}_scores_logists"])
            inst.boxes_sigma = torch.from_numpy(z[f"pseudo{i}_boxes_sigma"])
            r["instances"] = inst
        sampling.set_key_source(perm_key_source(opt.SeededPerm(93)))
        lu, _, _, _ = model(strong, branch="unsupervised", danchor=True)
        for k, v in lu.items():
            # the RPN terms (this test's subject: no weight on the unsupervised ones) at 1e-4 of the REAL reference's numbers.
            # The ROI terms are means over each side's OWN proposal sample from random-init, near-tied scores: one pair of
            # proposals swapping rank -- any 1e-5 change of the features does it; round 6's position-split convolution kernel
            # did, round 5's did not -- moves the position-indexed 512-ROI sample, `loss_cls` by ~1e-3 and the NLL
            # `loss_box_reg` (~16 at random init) by ~1e-2.  Against the golden they can only state "the same quantity on
            # another sample" (5e-2); the 1e-4 statement for them is made below on IDENTICAL proposals.
            close(v.detach().cpu(), z["unsup_" + k], 1e-4 if k.startswith("loss_rpn") else 5e-2, 1e-6, "unsup " + k)
        # ... and on identical proposals: the oracle (its `rpn_loss_weight` path is pinned to the same golden on CPU,
        # tests/test_oracle_golden.py) runs the same branch with its proposal stage answered by the HIP proposals
        # (tests/test_baseline_size_gpu.py::_ProposalLog, whose `check` also proves the HIP stage exact on its own inputs)
        from tests.test_baseline_size_gpu import _ProposalLog
        ocfg = opt.Cfg(num_classes=K, tau=tau, rpn_loss_weight=float(z["loss_weight"]),
                       rpn_bbox_reg_loss_weight=float(z["bbox_reg_loss_weight"]))
        oparams = opt.golden_params(opt.Cfg(num_classes=K, tau=tau), int(z["seed"]))
        ostrong = records(z, "strong", 2)
        for i, r in enumerate(ostrong):
            inst = opt.FreeInstances(tuple(r["image"].shape[-2:]))
            inst.pseudo_boxes = d2.Boxes(torch.from_numpy(z[f"pseudo{i}_pseudo_boxes"]))
            inst.scores_logists = torch.from_numpy(z[f"pseudo{i}_scores_logists"])
            inst.boxes_sigma = torch.from_numpy(z[f"pseudo{i}_boxes_sigma"])
            r["instances"] = inst
        with pytest.MonkeyPatch.context() as mp:
            log = _ProposalLog(mp, ocfg)
            sampling.set_key_source(perm_key_source(opt.SeededPerm(93)))
            lh, _, _, _ = model(strong, branch="unsupervised", danchor=True)
            lo, _, _, _ = opt.model_forward(ocfg, oparams, ostrong, "unsupervised", danchor=True, perm_fn=opt.SeededPerm(93))
            print(f"\n[rpn loss weight, unsupervised branch on identical proposals] {log.check()}")
        for k, v in lo.items():
            close(lh[k].detach().cpu(), v.detach(), 1e-4, 1e-6, "unsup (identical proposals) " + k)
    finally:
        sampling.set_key_source(None)



def test_trainer_resize_equals_the_reference_outputs_bit_for_bit():
    """PTrainer.resize (trainer.py:557-590) as the step calls it -- the whole list in one image launch, the boxes of all records
    in ONE multiply and ONE add over their concatenation -- against what the REAL reference's PTrainer.resize produced for the
    same records and ratios (tests/golden/trainer_pieces.npz): uint8 canvases, gt boxes and pseudo boxes bit for bit; untouched
    fields pass through; and against the record-by-record fp32 statement on a larger mixed batch."""
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    z = load("trainer_pieces")
    recs = records(z, "rz_in", 2, FreeInstances)
    recs[1]["instances"].pseudo_boxes = Boxes(torch.from_numpy(z["rz_in1_pseudo_boxes"]))
    cfg = setup_cfg("configs/pt/final_c2f.yaml", ["MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", ""])
    ratios = [float(q) for q in z["rz_ratios"]]
    tr = PTrainer(cfg, ratio_fn=lambda: ratios.pop(0))
    out = tr.resize([dict(r) for r in recs])
    for i, o in enumerate(out):
        assert np.array_equal(o["image"].cpu().numpy(), z[f"rz_out{i}_image"]), f"canvas {i}"
        assert np.array_equal(o["instances"].gt_boxes.tensor.cpu().numpy(), z[f"rz_out{i}_gt_boxes"]), f"gt boxes {i}"
        assert torch.equal(o["instances"].gt_classes.cpu(), recs[i]["instances"].gt_classes.cpu())
    assert np.array_equal(out[1]["instances"].pseudo_boxes.tensor.cpu().numpy(), z["rz_out1_pseudo_boxes"]), "pseudo boxes"
    assert np.array_equal(recs[0]["instances"].gt_boxes.tensor.cpu().numpy(), z["rz_in0_gt_boxes"]), "the inputs are not modified"
    # a larger mixed batch (records without boxes, with one field, with both; empty box sets) against the in-place statement
    gen = torch.Generator().manual_seed(3)
    big, want, rr = [], [], []
    for i in range(9):
        inst = FreeInstances((120, 160))
        fields = {}
        if i % 3 != 0:
            fields["gt_boxes"] = torch.rand(0 if i == 4 else 5 + i, 4, generator=gen) * 150
        if i % 2 == 0:
            fields["pseudo_boxes"] = torch.rand(3 + i, 4, generator=gen) * 150
        for k, v in fields.items():
            inst.set(k, Boxes(v.clone()))
        inst.set("tag", torch.full((1,), float(i)))
        big.append({"image": torch.randint(0, 256, (3, 120, 160), generator=gen, dtype=torch.uint8), "instances": inst})
        ratio = 0.5 + 0.05 * i
        rr.append(ratio)
        dh, dw = int(120 * ratio), int(160 * ratio)
        x1, y1 = int((160 - dw) / 2), int((120 - dh) / 2)
        w_ = {}
        for k, v in fields.items():
            t = v.clone()
            t *= ratio
            t[:, 0] += x1
            t[:, 2] += x1
            t[:, 1] += y1
            t[:, 3] += y1
            w_[k] = t
        want.append(w_)
    tr._ratio_fn = lambda: rr.pop(0)
    got = tr.resize(big)
    for o, w_, src in zip(got, want, big):
        assert set(o["instances"].get_fields()) == set(src["instances"].get_fields())
        for k, t in w_.items():
            assert torch.equal(o["instances"].get(k).tensor.cpu(), t), k
        assert torch.equal(o["instances"].get("tag"), src["instances"].get("tag"))
