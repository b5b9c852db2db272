"""The oracle restatement (oracle/pt.py) against golden vectors produced by the REAL reference
(tools/gen_golden.py, dev container).  CPU only.

Tolerances: integers/indices exact; fp32 values rtol 2e-5 (the restatement uses the same torch
CPU ops as the reference run that produced the fixture, but not always in the same order)."""
import numpy as np
import pytest
import torch

from oracle import d2, pt as opt
from tests.helpers import close, load, records

RT, AT = 2e-5, 1e-6


def test_box_codec_matches_reference():
    z = load("box_codec")
    src, tgt = torch.from_numpy(z["src"]), torch.from_numpy(z["tgt"])
    for tag, w in (("rpn", (1.0, 1.0, 1.0, 1.0)), ("roi", (10.0, 10.0, 5.0, 5.0))):
        d = opt.get_deltas(src, tgt, w)
        close(d, z[f"deltas_{tag}"], RT, AT, f"get_deltas {tag}")
        out = opt.apply_deltas(torch.from_numpy(z[f"apply_in_{tag}"]), src, w)
        close(out, z[f"apply_out_{tag}"], RT, 1e-4, f"apply_deltas {tag}")
        close(opt.apply_deltas(d, src, w), z[f"roundtrip_{tag}"], RT, 1e-3, f"roundtrip {tag}")
    ka = opt.get_deltas(torch.from_numpy(z["ka_src"]), torch.from_numpy(z["ka_tgt"]), (10.0, 10.0, 5.0, 5.0))
    close(ka, z["ka_deltas"], RT, AT, "known answer")
    # hand-checkable known answers
    close(ka, [[-0.4000, -0.3750, -0.6392, -0.6677], [0.0, 1.8750, -1.1157, 0.5889]], 0, 6e-5, "KA table")
    pdf = opt.gaussian_dist_pdf(torch.from_numpy(z["pdf_val"]), torch.from_numpy(z["pdf_mean"]),
                                torch.from_numpy(z["pdf_var"]))
    close(pdf, z["pdf"], RT, 1e-9, "gaussian pdf")
    close(opt.gaussian_dist_pdf(torch.tensor([0.1, 0.5]), torch.tensor([0.0, 0.0]), torch.tensor([0.5, 0.2])),
          [0.4416, 0.3020], 0, 6e-5, "pdf KA")


def test_trainer_pieces_match_reference():
    z = load("trainer_pieces")
    cfg = opt.Cfg()
    recs = records(z, "rz_in", 2)
    recs[1]["instances"].pseudo_boxes = d2.Boxes(torch.from_numpy(z["rz_in1_pseudo_boxes"]))
    for i, (r, q) in enumerate(zip(recs, z["rz_ratios"])):
        out = opt.shrink_paste(cfg, r, float(q))
        assert np.array_equal(out["image"].numpy(), z[f"rz_out{i}_image"]), "resize image must be bit-exact"
        # the C restatement of ATen's evaluation order (what the HIP kernel is checked against at full size, on hosts
        # whose torch build / thread count may round differently) reproduces the real reference's bytes as well
        h, w = r["image"].shape[-2:]
        dh, dw = int(h * float(q)), int(w * float(q))
        x1, y1 = int((w - dw) / 2), int((h - dh) / 2)
        assert np.array_equal(d2.bilinear_shrink_u8(r["image"], dh, dw).numpy(),
                              z[f"rz_out{i}_image"][:, y1:y1 + dh, x1:x1 + dw]), "ptref_bilinear_shrink_u8 vs reference"
        close(out["instances"].gt_boxes.tensor, z[f"rz_out{i}_gt_boxes"], 1e-6, 1e-5, "resize boxes")
    close(out["instances"].pseudo_boxes.tensor, z["rz_out1_pseudo_boxes"], 1e-6, 1e-5, "resize pseudo boxes")
    # ... and equals F.interpolate itself on this host (multi-threaded generic ATen kernel) at BASELINE size
    if torch.get_num_threads() > 1:
        import torch.nn.functional as F
        img = torch.from_numpy(np.random.RandomState(5).randint(0, 256, (3, 800, 1333)).astype(np.uint8))
        for ratio in (0.5, 0.731, 0.9999, 1.0):
            dh, dw = int(800 * ratio), int(1333 * ratio)
            ref = torch.zeros((3, dh, dw), dtype=torch.uint8)
            ref[:] = F.interpolate(img.unsqueeze(0).float(), size=(dh, dw), align_corners=False, mode="bilinear")[0]
            assert torch.equal(d2.bilinear_shrink_u8(img, dh, dw), ref), f"ratio {ratio}"
    # inputs of shrink_paste are not mutated
    assert np.array_equal(recs[0]["instances"].gt_boxes.tensor.numpy(), z["rz_in0_gt_boxes"])
    # EMA
    s, t = torch.from_numpy(z["ema_s"]), torch.from_numpy(z["ema_t"])
    out = opt.ema_update({"p": s}, {"p": t.clone()}, 0.9996)["p"]
    assert np.array_equal(out.numpy(), z["ema_out"]), "EMA must be bit-exact (same op order)"
    # clip
    g = torch.from_numpy(z["clip_in"]).clone()
    total = opt.clip_gradient([g], 10.0)
    close(g, z["clip_out"], 2e-6, 1e-9, "clip")
    assert total > 10.0 and abs(float(g.norm()) - 10.0) < 1e-4


def _ocfg(z, anchor, burn=1):
    return opt.Cfg(num_classes=int(z["K"]), anchor_generator=anchor, tau=tuple(float(v) for v in z["tau"]),
                   burn_up_step=burn)


def _grad_check(z, prefix, params, names):
    for k in names:
        key = f"{prefix}_norm_{k}"
        if key not in z.files:
            continue
        g = params[k].grad
        close(g.double().norm(), z[key], 2e-4, 1e-7, key)
        close(g.double().sum(), z[f"{prefix}_sum_{k}"], 2e-3, 2e-4 * float(z[key]) + 1e-7, f"{prefix}_sum_{k}")
        close(g.flatten()[:32], z[f"{prefix}_head_{k}"], 2e-3, 1e-5 * float(z[key]) + 1e-8, f"{prefix}_head_{k}")


@pytest.mark.parametrize("anchor,tag", [("DefaultAnchorGenerator", "default_anchor"),
                                        ("DifferentiableAnchorGenerator", "diff_anchor")])
def test_model_branches_match_reference(anchor, tag):
    z = load("model_" + tag)
    cfg = _ocfg(z, anchor)
    params = opt.golden_params(cfg, int(z["seed"]))
    names = opt.trainable_names(cfg, params)
    for n in names:
        params[n].requires_grad_(True)

    # supervised
    recs = records(z, "sup", 2)
    perm = opt.SeededPerm(77)
    losses, _, _, _ = opt.model_forward(cfg, params, recs, "supervised", perm_fn=perm)
    assert perm.log == list(z["sup_perm_log"]), "sampler draw sizes (== label counts) must match exactly"
    for k, v in losses.items():
        close(v.detach(), z["sup_" + k], RT, AT, "sup " + k)
    sum(losses.values()).backward()
    _grad_check(z, "supgrad", params, names)

    # teacher
    weak = records(z, "weak", 2)
    with torch.no_grad():
        _, prop_rpn, prop_roih, pred = opt.model_forward(cfg, params, weak, "unsup_data_weak", perm_fn=opt.SeededPerm(78))
    for i in range(2):
        close(prop_rpn[i].proposal_boxes.tensor, z[f"t_rpn{i}_proposal_boxes"], RT, 2e-4, "rpn boxes")
        close(prop_rpn[i].objectness_logits, z[f"t_rpn{i}_objectness_logits"], RT, 1e-5, "rpn logits")
        assert np.array_equal(prop_roih[i].pred_classes.numpy(), z[f"t_roih{i}_pred_classes"])
        close(prop_roih[i].pred_boxes.tensor, z[f"t_roih{i}_pred_boxes"], RT, 2e-4, "det boxes")
        close(prop_roih[i].scores, z[f"t_roih{i}_scores"], RT, 1e-6, "det scores")
        close(prop_roih[i].scores_logists, z[f"t_roih{i}_scores_logists"], RT, 1e-5, "det logits")
        close(prop_roih[i].boxes_sigma, z[f"t_roih{i}_boxes_sigma"], RT, 1e-5, "det sigma")
    close(pred[0], z["t_pred_scores"], RT, 1e-5, "roi scores")
    close(pred[1], z["t_pred_deltas"], RT, 1e-5, "roi deltas")

    # unsupervised (student on the teacher's pseudo labels, danchor=True)
    strong = records(z, "strong", 2)
    pseudo = opt.pseudo_labels_from_teacher(prop_roih)
    for r, p in zip(strong, pseudo):
        r["instances"] = p
    for n in names:
        params[n].grad = None
    losses_u, _, _, _ = opt.model_forward(cfg, params, strong, "unsupervised", danchor=True, perm_fn=opt.SeededPerm(79))
    for k, v in losses_u.items():
        close(v.detach(), z["unsup_" + k], RT, AT, "unsup " + k)
    sum(losses_u.values()).backward()
    _grad_check(z, "unsupgrad", params, names)
    if anchor == "DifferentiableAnchorGenerator":
        assert "unsupgrad_norm_proposal_generator.anchor_generator.anchor_0" in z.files
        assert float(z["unsupgrad_norm_proposal_generator.anchor_generator.anchor_0"]) > 0


def test_run_step_matches_reference():
    """Three real PTrainer.run_step iterations: burn-in, EMA copy + mutual learning, EMA + mutual."""
    z = load("run_step")
    cfg = _ocfg(z, "DifferentiableAnchorGenerator", burn=1)
    cfg.tau = tuple(float(v) for v in z["tau"])
    state = {"student": opt.golden_params(cfg, int(z["seed"])), "teacher": opt.golden_params(cfg, int(z["teacher_seed"])),
             "bufs": {}, "iter": 0}
    B = int(z["B"])
    probes = sorted({k.split("_s_sum_")[1] for k in z.files if "_s_sum_" in k})
    for it in range(3):
        data = tuple(records(z, f"it{it}_{nm}", B) for nm in ("lq", "lk", "uq", "uk"))
        rr = [float(v) for v in z[f"it{it}_ratios"]]
        if it < cfg.burn_up_step:
            ratios = {"label": rr, "unlabel": []}
        else:   # reference order: resize(unlabel_q) first, then resize(label_q)  (trainer.py:333-334)
            ratios = {"unlabel": rr[:B], "label": rr[B:2 * B]}
        override = None
        if f"it{it}_pseudo0_pseudo_boxes" in z.files:
            override = []
            for i in range(B):
                h, w = data[3][i]["image"].shape[-2:]
                inst = opt.FreeInstances((h, w))
                inst.pseudo_boxes = d2.Boxes(torch.from_numpy(z[f"it{it}_pseudo{i}_pseudo_boxes"]))
                inst.scores_logists = torch.from_numpy(z[f"it{it}_pseudo{i}_scores_logists"])
                inst.boxes_sigma = torch.from_numpy(z[f"it{it}_pseudo{i}_boxes_sigma"])
                override.append(inst)
        m = opt.run_step(cfg, state, data, ratios, perm_fn=opt.SeededPerm(500 + it), pseudo_override=override)
        if override is not None:
            # the oracle teacher's own pseudo labels vs the reference's (weights differ by fp32 noise after a
            # real optimiser step, so tolerance not bit-exactness)
            for mine, ref in zip(state["last_pseudo"], override):
                close(mine.pseudo_boxes.tensor, ref.pseudo_boxes.tensor, 1e-4, 5e-3, "pseudo boxes")
                close(mine.scores_logists, ref.scores_logists, 1e-4, 1e-3, "pseudo logits")
                close(mine.boxes_sigma, ref.boxes_sigma, 1e-4, 1e-3, "pseudo sigma")
        for k in z.files:
            if k.startswith(f"it{it}_m_"):
                close(m[k[len(f"it{it}_m_"):]], z[k], 2e-4, 1e-6, k)
        for k in probes:
            close(state["student"][k].double().sum(), z[f"it{it}_s_sum_{k}"], 1e-5, 1e-4, f"student sum {k}")
            close(state["student"][k].flatten()[:16], z[f"it{it}_s_head_{k}"], 1e-5, 1e-7, f"student head {k}")
            close(state["teacher"][k].double().sum(), z[f"it{it}_t_sum_{k}"], 1e-5, 1e-4, f"teacher sum {k}")
            close(state["teacher"][k].flatten()[:16], z[f"it{it}_t_head_{k}"], 1e-5, 1e-7, f"teacher head {k}")


def test_run_step_long_matches_reference_over_14_iterations():
    """Round 6: FOURTEEN real PTrainer.run_step iterations of the reference (tools/gen_golden.py::
    gen_run_step_long: 3 burn-in, the keep_rate = 0 copy step, 10 EMA-0.9996 mutual-learning steps, the reference's LR warm-up
    active) replayed by the oracle from the same start: the reference's MULTI-STEP dynamics -- momentum accumulating over 14 steps,
    weight decay, the teacher drifting away from the student by EMA, pseudo labels of a changing teacher, the warm-up schedule --
    pin the oracle's, per iteration: every metric, 7 parameter probes of student and teacher, the teacher's pseudo labels."""
    z = load("run_step_long")
    burn, iters = int(z["burn"]), int(z["iters"])
    assert iters >= 12 and iters - burn - 1 >= 8
    cfg = _ocfg(z, "DifferentiableAnchorGenerator", burn=burn)
    cfg.tau = tuple(float(v) for v in z["tau"])
    assert cfg.base_lr == float(z["base_lr"]) and cfg.warmup_iters == int(z["warmup_iters"]) and cfg.ema_keep_rate == float(z["ema_keep_rate"])
    state = {"student": opt.golden_params(cfg, int(z["seed"])), "teacher": opt.golden_params(cfg, int(z["teacher_seed"])),
             "bufs": {}, "iter": 0}
    B = int(z["B"])
    probes = sorted({k.split("_s_sum_")[1] for k in z.files if "_s_sum_" in k})
    live = 0
    for it in range(iters):
        assert abs(opt.lr_at(cfg, it) - float(z[f"it{it}_lr"])) <= 1e-12, "LR schedule"
        data = tuple(records(z, f"it{it}_{nm}", B) for nm in ("lq", "lk", "uq", "uk"))
        rr = [float(v) for v in z[f"it{it}_ratios"]]
        ratios = {"label": rr, "unlabel": []} if it < burn else {"unlabel": rr[:B], "label": rr[B:2 * B]}
        override = None
        if f"it{it}_pseudo0_pseudo_boxes" in z.files:
            override = []
            for i in range(B):
                h, w = data[3][i]["image"].shape[-2:]
                inst = opt.FreeInstances((h, w))
                inst.pseudo_boxes = d2.Boxes(torch.from_numpy(z[f"it{it}_pseudo{i}_pseudo_boxes"]))
                inst.scores_logists = torch.from_numpy(z[f"it{it}_pseudo{i}_scores_logists"])
                inst.boxes_sigma = torch.from_numpy(z[f"it{it}_pseudo{i}_boxes_sigma"])
                override.append(inst)
        assert (override is not None) == (it >= burn)
        m = opt.run_step(cfg, state, data, ratios, perm_fn=opt.SeededPerm(700 + it), pseudo_override=override)
        if override is not None:
            for mine, ref in zip(state["last_pseudo"], override):
                assert len(ref) > 0
                close(mine.pseudo_boxes.tensor, ref.pseudo_boxes.tensor, 1e-4, 5e-3, f"it {it} pseudo boxes")
                close(mine.scores_logists, ref.scores_logists, 1e-4, 1e-3, f"it {it} pseudo logits")
                close(mine.boxes_sigma, ref.boxes_sigma, 1e-4, 1e-3, f"it {it} pseudo sigma")
        for k in z.files:
            if k.startswith(f"it{it}_m_"):
                name = k[len(f"it{it}_m_"):]
                if np.isnan(z[k]):           # the reference's own empty-mean terms (the copy step's ROI terms): NaN on both sides
                    assert np.isnan(m[name]), f"{k}: reference NaN, oracle {m[name]}"
                    continue
                close(m[name], z[k], 2e-4, 1e-6, k)
                live += name.endswith("_unsup") and abs(float(z[k])) > 1e-6
        for k in probes:
            # sums of up to 25.7 M weights that cancel to ~0.4 (fc1): the per-weight fp32 differences of two summation orders add up
            # like a random walk -- 1e-4 absolute over the first iterations (as the 3-iteration test), 5e-4 once 14 optimiser steps
            # have compounded them (measured 1.7e-4 at iteration 7); the 16-value heads below are the sharp check
            sum_atol = 1e-4 if it < 3 else 5e-4
            close(state["student"][k].double().sum(), z[f"it{it}_s_sum_{k}"], 1e-5, sum_atol, f"it {it} student sum {k}")
            close(state["student"][k].flatten()[:16], z[f"it{it}_s_head_{k}"], 1e-5, 1e-7, f"it {it} student head {k}")
            close(state["teacher"][k].double().sum(), z[f"it{it}_t_sum_{k}"], 1e-5, sum_atol, f"it {it} teacher sum {k}")
            close(state["teacher"][k].flatten()[:16], z[f"it{it}_t_head_{k}"], 1e-5, 1e-7, f"it {it} teacher head {k}")
    assert live >= 4 * (iters - burn - 1), f"only {live} live unsupervised loss values in the fixture"
    # the EMA teacher really moved away from the student (else the last ten steps would pin nothing the copy step does not)
    k = "roi_heads.box_predictor.cls_score.bias"
    assert float((state["student"][k] - state["teacher"][k]).abs().max()) > 1e-4


def test_rpn_loss_weight_is_applied_twice_to_supervised_losses_only():
    """MODEL.RPN.LOSS_WEIGHT = 2, BBOX_REG_LOSS_WEIGHT = 0.5 on the REAL reference model (tests/golden/rpn_loss_weight.npz):
    the supervised RPN losses carry the weight dict squared (rpn.py:254 and rpn.py:141), the unsupervised ones no weight
    (rpn.py:347-360); the ROI losses are untouched."""
    z = load("rpn_loss_weight")
    cfg = opt.Cfg(num_classes=int(z["K"]), tau=tuple(float(v) for v in z["tau"]), rpn_loss_weight=float(z["loss_weight"]),
                  rpn_bbox_reg_loss_weight=float(z["bbox_reg_loss_weight"]))
    params = opt.golden_params(cfg, int(z["seed"]))
    losses, _, _, _ = opt.model_forward(cfg, params, records(z, "sup", 2), "supervised", perm_fn=opt.SeededPerm(91))
    for k, v in losses.items():
        close(v.detach(), z["sup_" + k], RT, AT, "sup " + k)
    # the same model with unit weights: cls x 2^2, loc x (0.5 * 2)^2
    unit = opt.Cfg(num_classes=int(z["K"]), tau=tuple(float(v) for v in z["tau"]))
    l1, _, _, _ = opt.model_forward(unit, params, records(z, "sup", 2), "supervised", perm_fn=opt.SeededPerm(91))
    close(losses["loss_rpn_cls"].detach(), 4.0 * l1["loss_rpn_cls"].detach(), 1e-6, 0, "cls weight squared")
    close(losses["loss_rpn_loc"].detach(), 1.0 * l1["loss_rpn_loc"].detach(), 1e-6, 0, "loc weight squared")
    strong = records(z, "strong", 2)
    for i, r in enumerate(strong):
        inst = opt.FreeInstances(tuple(r["image"].shape[-2:]))
        inst.pseudo_boxes = d2.Boxes(torch.from_numpy(z[f"pseudo{i}_pseudo_boxes"]))
        inst.scores_logists = torch.from_numpy(z[f"pseudo{i}_scores_logists"])
        inst.boxes_sigma = torch.from_numpy(z[f"pseudo{i}_boxes_sigma"])
        r["instances"] = inst
    lu, _, _, _ = opt.model_forward(cfg, params, strong, "unsupervised", danchor=True, perm_fn=opt.SeededPerm(93))
    for k, v in lu.items():
        close(v.detach(), z["unsup_" + k], RT, AT, "unsup " + k)

