"""PTrainer: the teacher+student train step on MI355X (reference pt/engine/trainer.py).

`run_step` follows trainer.py:263-392 call for call -- burn-in vs mutual learning, teacher pseudo-labelling
without thresholding, shrink-and-paste `resize`, supervised + unsupervised student forwards, one backward,
gradient clipping, SGD -- but every heavy piece is a HIP kernel and the per-tensor Python loops of the reference
(EMA, clip, SGD, metrics `.item()`s) are single launches over flat buffers.  What the reference does only to burn
time (anomaly mode, empty_cache, gc.collect, a third unused model copy) is not replicated."""
import random
import time
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist

from .. import ops
from ..modeling import EnsembleTSModel, build_model
from ..structures import Boxes, FreeInstances
from ..solver import check_optimizer_options, lr_at
from .flat import BucketedGradReducer, FlatParams, broadcast_


class PTrainer:
    @staticmethod
    def ddp_wgrad_waves(amp: bool) -> int:
        """weight-gradient waves PTrainer selects while a gradient exchange is active"""
        return 2 if amp else 3

    def __init__(self, cfg, data_loader=None, ratio_fn: Optional[Callable[[], float]] = None,
                 force_grad_reducer: bool = False, grad_reduce: str = "all_reduce"):
        """force_grad_reducer: run the bucketed gradient all-reduce (hooks + collectives) even with one rank -- needs an
        initialised process group; the sum over one rank is the identity (single-GPU validation of the DDP path).
        grad_reduce: "all_reduce" or "reduce_scatter" (engine/flat.py: BucketedGradReducer)."""
        self.cfg = cfg
        check_optimizer_options(cfg)
        # reference trainer.py:98 (cfg.SOLVER.AMP.ENABLED): mixed precision for the conv / FC GEMMs (see ops.py).  The mode
        # belongs to THIS trainer and is applied around each of its steps (ops.operand_rounding): "bf16", None, or -- for
        # the comparison runs of tools/ and tests/ -- "bf16_emulate"
        self.operand_rounding = "bf16" if cfg.SOLVER.AMP.ENABLED else None
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.model = build_model(cfg)                  # student
        self.model_teacher = build_model(cfg)          # teacher (per-rank replica, never all-reduced)
        self.model.train()
        self.model_teacher.train()                     # the reference never puts the teacher in eval mode (:302-303)
        for p in self.model_teacher.parameters():
            p.requires_grad_(False)
        self.student = FlatParams(self.model)
        # teacher must share the student's flat layout: order by the student's index
        self.teacher = _flatten_like(self.model_teacher, self.student)
        broadcast_(self.student.flat)                  # trainer.py:495 _sync_params_and_buffers
        broadcast_(self.teacher.flat)
        self.momentum_buf = torch.zeros_like(self.student.trainable())
        # gradient exchange overlapped with backward: 16 MB buckets from the tail of the flat buffer (box head first)
        self.reducer = BucketedGradReducer(self.student, self.world_size, bucket_elems=4 * 1024 * 1024,
                                           force=force_grad_reducer, mode=grad_reduce)
        # CU sharing with the collectives (tools/exp/contention.py): with the gradient exchange active its kernels hold CUs
        # while backward runs -- the persistent convolution then draws its tiles from work queues, and the weight-gradient kernels
        # launch several waves of shorter workgroups, so that a held CU costs its share instead of a second pass.  On one GPU the
        # static walk / one workgroup per CU is 0.5 - 3 % faster.  Waves chosen by profiles/r06_contention_waves_1_to_4.txt /
        # r06_contention_amp_waves_1_to_4.txt: fp32 3 (+1.4 % without contention, x1.13 with 8 CUs held; 4 costs +2.5 % for x1.10),
        # bf16-storage kernels 2 (+1.2 %, and no worse under contention than 3 or 4)
        ops.set_tile_schedule("dynamic" if self.reducer.active else "static")
        ops.set_wgrad_waves(self.ddp_wgrad_waves(bool(cfg.SOLVER.AMP.ENABLED)) if self.reducer.active else 1)
        ops.set_p8_conv_waves(16 if self.reducer.active else 1)       # (SOLVER.AMP.ENABLED: the bf16-storage convolution, persistent too)
        self._first_step = True
        self.joint_student_pass = True      # one backbone pass for the two student branches when they share a canvas
        self.ensem_ts_model = EnsembleTSModel(self.model_teacher, self.model)
        self.iter = self.start_iter = 0
        self.max_iter = cfg.SOLVER.MAX_ITER
        self._data_iter = iter(data_loader) if data_loader is not None else None
        self._ratio_fn = ratio_fn or (lambda: random.uniform(0.5, 1.0))
        self.last_metrics: Dict[str, float] = {}
        self._mean_int = [int(m) for m in cfg.MODEL.PIXEL_MEAN]     # pixel_mean.cpu().int() (trainer.py:569)

    # ------------------------------------------------------------------ pseudo-labelling (trainer.py:179-257)
    def threshold_bbox(self, inst, proposal_type="roih"):
        new = FreeInstances(inst.image_size)
        if proposal_type == "rpn":
            new.gt_boxes = Boxes(inst.proposal_boxes.tensor)
            new.objectness_logits = inst.objectness_logits
            new.pseudo_boxes = Boxes(inst.proposal_boxes.tensor)
        elif proposal_type == "roih":
            new.pseudo_boxes = Boxes(inst.pred_boxes.tensor)
            new.scores_logists = inst.scores_logists
            if inst.has("boxes_sigma"):
                new.boxes_sigma = inst.boxes_sigma
        return new

    def process_pseudo_label(self, proposals, proposal_type, psedo_label_method=""):
        if psedo_label_method != "all":
            raise ValueError("Unkown pseudo label boxes methods")
        out = [self.threshold_bbox(p, proposal_type) for p in proposals]
        n = sum(len(p) for p in out) / max(len(out), 1)
        return out, n

    @staticmethod
    def remove_label(data):
        for d in data:
            d.pop("instances", None)
        return data

    @staticmethod
    def add_label(data, labels):
        for d, lab in zip(data, labels):
            d["instances"] = lab
        return data

    # ------------------------------------------------------------------ resize (trainer.py:557-590)
    def resize(self, data: List[dict]) -> List[dict]:
        """One launch for the images of the whole list (ops.shrink_paste_batch); boxes are rescaled with the same
        three in-place fp32 steps as the reference (`*= ratio`, `+= x1`, `+= y1`)."""
        dev = self.model.device
        images = [rec["image"].to(dev, non_blocking=True) for rec in data]
        ratios = [self._ratio_fn() for _ in data]
        canvases, offsets = ops.shrink_paste_batch(images, ratios, self._mean_int)
        # boxes of the whole list: ONE multiply and ONE add over the concatenation (per element exactly the reference's fp32
        # `*= ratio` followed by `+= x1` / `+= y1`; a clone + five in-place launches per record were 6 n tiny launches)
        box_fields = [(i, k, v.tensor) for i, rec in enumerate(data) for k, v in rec["instances"].get_fields().items()
                      if k in ("gt_boxes", "pseudo_boxes")]
        scaled = {}
        if box_fields:
            counts = [int(t.shape[0]) for _, _, t in box_fields]
            allb = torch.cat([t.to(dev) for _, _, t in box_fields], 0) if len(box_fields) > 1 else box_fields[0][2].to(dev).clone()
            rows = torch.tensor([[ratios[i], float(offsets[i][0]), float(offsets[i][1]), float(offsets[i][0]), float(offsets[i][1])]
                                 for i, _, _ in box_fields], dtype=torch.float32)
            if dev.type == "cuda":             # per-record rows travel (small, pinned, asynchronous); the per-box expansion runs on the device
                rows = torch.repeat_interleave(rows.pin_memory().to(dev, non_blocking=True), ops.dev_i32(counts, dev).long(), dim=0,
                                               output_size=sum(counts))
            else:
                rows = torch.repeat_interleave(rows, torch.tensor(counts), dim=0).to(dev)
            allb *= rows[:, 0:1]
            allb += rows[:, 1:5]
            c0 = 0
            for (i, k, _), c in zip(box_fields, counts):
                scaled[(i, k)] = allb[c0:c0 + c]
                c0 += c
        out = []
        for i, (rec, canvas) in enumerate(zip(data, canvases)):
            new = dict(rec)
            new["image"] = canvas
            inst = rec["instances"]
            ni = FreeInstances(inst.image_size)
            for k, v in inst.get_fields().items():
                ni.set(k, Boxes(scaled[(i, k)]) if (i, k) in scaled else v)
            new["instances"] = ni
            out.append(new)
        return out

    # ------------------------------------------------------------------ EMA / optimiser
    @torch.no_grad()
    def _update_teacher_model(self, keep_rate=0.996):
        """trainer.py:431-449 as ONE launch over the flat buffers: t = s*(1-k) + t*k."""
        ops.ema_update(self.student.flat, self.teacher.flat, keep_rate)

    @torch.no_grad()
    def _clip_and_step(self, clip_norm: float):
        """trainer.py:385-386: clip_gradient(model, 10.) + optimizer.step() fused: one reduction + one update."""
        g = self.student.grad
        ss = ops.sumsq(g)
        ops.clip_sgd_step(self.student.trainable(), g, self.momentum_buf, ss, clip_norm, lr_at(self.cfg, self.iter),
                          self.cfg.SOLVER.MOMENTUM, self.cfg.SOLVER.WEIGHT_DECAY, self._first_step)
        self._first_step = False
        return ss

    # ------------------------------------------------------------------ the step (trainer.py:263-392)
    def run_step(self, data=None) -> Dict[str, float]:
        with ops.operand_rounding(self.operand_rounding):
            return self._run_step(data)

    def _run_step(self, data=None) -> Dict[str, float]:
        assert self.model.training, "[PTrainer] model was changed to eval mode!"
        start = time.perf_counter()
        if data is None:
            data = next(self._data_iter)
        label_data_q, label_data_k, unlabel_data_q, unlabel_data_k = [list(d) for d in data]
        data_time = time.perf_counter() - start
        U = self.cfg.UNSUPNET
        self.student.zero_grad()

        if self.iter < U.BURN_UP_STEP:
            batch = self.resize(label_data_q + label_data_k)
            record_dict, _, _, _ = self.model(batch, branch="supervised")
            losses = sum(v * 1.0 for k, v in record_dict.items() if k[:4] == "loss")
        else:
            if self.iter == U.BURN_UP_STEP:
                self._update_teacher_model(keep_rate=0.00)
            elif (self.iter - U.BURN_UP_STEP) % U.TEACHER_UPDATE_ITER == 0:
                self._update_teacher_model(keep_rate=U.EMA_KEEP_RATE)
            with torch.no_grad():
                _, _, proposals_roih_unsup_k, _ = self.model_teacher(unlabel_data_k, branch="unsup_data_weak")
            pseudo, _ = self.process_pseudo_label(proposals_roih_unsup_k, "roih", "all")
            unlabel_data_q = self.add_label(self.remove_label([dict(d) for d in unlabel_data_q]), pseudo)
            unlabel_data_q = self.resize(unlabel_data_q)
            label_data_q = self.resize(label_data_q)
            record_dict = {}
            sup_batch = label_data_q + label_data_k
            if self.joint_student_pass and self.model.can_run_jointly(sup_batch, unlabel_data_q):
                # one backbone pass for both student branches (same canvas): fewer, fuller launches
                rec_sup, rec_unsup = self.model.forward_joint(sup_batch, unlabel_data_q, danchor=True)
            else:
                rec_sup, _, _, _ = self.model(sup_batch, branch="supervised")
                rec_unsup, _, _, _ = self.model(unlabel_data_q, branch="unsupervised", danchor=True)
            record_dict.update({k + "_sup": v for k, v in rec_sup.items()})
            record_dict.update({k + "_unsup": v for k, v in rec_unsup.items()})
            losses = 0
            for k, v in record_dict.items():
                if k[:4] != "loss":
                    continue
                tail = k.split("_")[-1]
                if tail == "sup":
                    losses = losses + v * U.SOURCE_LOSS_WEIGHT
                elif tail == "unsup":
                    losses = losses + v * U.TARGET_UNSUP_LOSS_WEIGHT
                else:
                    raise NotImplementedError

        losses.backward()
        self.reducer.finish()                  # DDP gradient average (buckets were launched during backward)
        ss = self._clip_and_step(10.0)
        self._write_metrics(record_dict, data_time, ss)
        self.iter += 1
        return self.last_metrics

    # every loss key a step can produce (burn-in: bare names; mutual learning: _sup / _unsup), in a fixed order, so that
    # ranks exchange equally laid-out vectors whatever subset each of them holds
    METRIC_KEYS = tuple(k + sfx for sfx in ("", "_sup", "_unsup")
                        for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"))

    def _write_metrics(self, record_dict, data_time, sumsq):
        """trainer.py:394-429 with ONE device->host copy: all loss scalars packed into a single tensor (the
        reference does ~10 `.cpu().item()` syncs and a pickled gloo gather).  Multi-rank semantics of the reference:
        the keys of rank 0, each averaged over ALL ranks with 0.0 for a rank that lacks the key (:413-417); data_time is
        the maximum over ranks (:407-411).  One all-gather of a fixed-layout vector (values + presence + data_time)."""
        extra = [k for k in record_dict if k not in self.METRIC_KEYS]
        if self.world_size > 1 and extra:
            raise KeyError(f"metrics {extra} have no slot in the cross-rank layout (PTrainer.METRIC_KEYS)")
        keys = list(self.METRIC_KEYS) + extra
        dev = sumsq.device
        zero = torch.zeros((), device=dev)
        present = [k in record_dict for k in keys]
        packed = torch.stack([record_dict[k].detach().float() if p else zero for k, p in zip(keys, present)] +
                             [sumsq.reshape(())])
        if self.world_size > 1:
            mine = torch.cat([packed, torch.tensor([float(p) for p in present] + [data_time], device=dev)])
            allv = torch.empty(self.world_size * mine.numel(), device=dev)
            dist.all_gather_into_tensor(allv, mine)
            allv = allv.view(self.world_size, mine.numel()).cpu()
            nk = len(keys)
            mean = allv[:, :nk].mean(dim=0).tolist()            # absent entries were packed as 0.0
            m = {k: v for k, v, p in zip(keys, mean, allv[0, nk + 1:2 * nk + 1].tolist()) if p > 0}   # rank 0's keys
            data_time = float(allv[:, -1].max())
            gsq = float(allv[dist.get_rank(), nk])
        else:
            vals = packed.cpu().tolist()
            m = {k: v for k, v, p in zip(keys, vals[:-1], present) if p}
            gsq = vals[-1]
        m["total_loss"] = sum(v for k, v in m.items() if k[:4] == "loss")
        # the fused ReLU (v_max_f32) turns a NaN pre-activation into 0 where torch.relu propagates it, so a diverged backbone
        # can hide from the loss values; the gradient norm (read back here anyway) is the cheap place where it still shows.
        # The reference runs under torch.autograd.set_detect_anomaly (trainer.py:266) and raises out of backward().
        if gsq != gsq or gsq in (float("inf"), float("-inf")):
            raise FloatingPointError(f"non-finite gradient norm at iteration {self.iter} (losses {m})")
        m["grad_norm"] = gsq ** 0.5
        m["data_time"] = data_time
        self.last_metrics = m

    # ------------------------------------------------------------------ evaluation (trainer.py:127-137, 529-542)
    @classmethod
    def build_evaluator(cls, cfg, class_names, is_2007: bool = False):
        """trainer.py:127-137: TEST.EVALUATOR selects the evaluator; the VOC protocol ("VOCeval", the README's mAP50
        tables) is implemented (probabilisticteacher_amd/evaluation.py); COCOeval needs pycocotools and is not."""
        from ..evaluation import PascalVOCDetectionEvaluator
        if cfg.TEST.EVALUATOR == "VOCeval":
            return PascalVOCDetectionEvaluator(class_names, is_2007=is_2007)
        if cfg.TEST.EVALUATOR == "COCOeval":
            raise NotImplementedError("TEST.EVALUATOR=COCOeval needs pycocotools; use VOCeval")
        raise ValueError("Unknown test evaluator.")

    @classmethod
    def build_train_loader(cls, cfg):
        """trainer.py:139-141 -> pt/data/build.py:107: the two-crop semi-supervised loader over cfg.DATASETS.TRAIN_LABEL /
        TRAIN_UNLABEL (device-side mapper, data/build.py)"""
        from ..data import build_detection_semisup_train_loader_two_crops
        return build_detection_semisup_train_loader_two_crops(cfg)

    @classmethod
    def build_test_loader(cls, cfg, dataset_name):
        from ..data import build_detection_test_loader
        return build_detection_test_loader(cfg, dataset_name)

    @classmethod
    def test(cls, cfg, model, data_loader=None, class_names=None, is_2007: bool = False):
        """DefaultTrainer.test(cfg, model): eval-mode inference + the evaluator over every dataset of cfg.DATASETS.TEST
        (results keyed by dataset name when there are several, flat for one -- D2's convention), or over an explicit
        `data_loader` of record batches with `class_names`.  Returns {"bbox": {"AP", "AP50", "AP75"}, ...}; on ranks other than
        0 of a multi-rank run the evaluator returns None (predictions are gathered on rank 0) and so does this."""
        from ..evaluation import inference_on_dataset
        amp = "bf16" if cfg.SOLVER.AMP.ENABLED else None
        if data_loader is not None:
            with ops.operand_rounding(amp):
                return inference_on_dataset(model, data_loader, cls.build_evaluator(cfg, class_names, is_2007))
        from ..data import datasets
        results = {}
        for name in cfg.DATASETS.TEST:
            meta = datasets.metadata(name)
            with ops.operand_rounding(amp):
                results[name] = inference_on_dataset(model, cls.build_test_loader(cfg, name),
                                                     cls.build_evaluator(cfg, meta["thing_classes"], meta["year"] == 2007))
        if len(results) == 1:
            results = list(results.values())[0]
        return results

    # ------------------------------------------------------------------ trainer shell (trainer.py:466-547)
    def resume_or_load(self, resume: bool = False):
        """trainer.py:466-496: weights (resume=False) or weights + optimiser + iteration (resume=True) from
        cfg.MODEL.WEIGHTS; parameters are then broadcast from rank 0 (`_sync_params_and_buffers`)."""
        from .. import checkpoint
        inc = checkpoint.resume_or_load(self, resume=resume)
        broadcast_(self.student.flat)
        broadcast_(self.teacher.flat)
        if self.world_size > 1:
            t = torch.tensor([self.start_iter], device=self.student.flat.device)
            dist.broadcast(t, src=0)
            self.iter = self.start_iter = int(t.item())
        return inc

    def _run_eval_hooks(self, eval_fn) -> Dict[str, float]:
        """trainer.py:529-542: `test_and_save_results_student` (keys suffixed `_student`) then `..._teacher`, each an EvalHook;
        results flattened the way D2's EvalHook does (`bbox_student/AP50`, `bbox/AP50`) for the metrics writer."""
        res_s = eval_fn(self.cfg, self.model)
        self._last_eval_results_student = res_s
        res_t = eval_fn(self.cfg, self.model_teacher)
        self._last_eval_results_teacher = res_t
        flat = {}
        for res, sfx in ((res_s, "_student"), (res_t, "")):
            for k, v in (res or {}).items():
                if isinstance(v, dict):
                    for kk, vv in v.items():
                        if isinstance(vv, (int, float)):
                            flat[f"{k}{sfx}/{kk}"] = float(vv)
        return flat

    def train(self, start_iter: Optional[int] = None, max_iter: Optional[int] = None, log_period: int = 20, eval_fn=None,
              run_eval: bool = True):
        """TrainerBase.train with the reference's hooks (trainer.py:498-547): LR schedule (folded into the fused step),
        PeriodicCheckpointer (rank 0: model_{iter:07d}.pth every CHECKPOINT_PERIOD iterations, model_final.pth,
        `last_checkpoint`), the two EvalHooks (student, then teacher, every TEST.EVAL_PERIOD iterations and after the last
        one; `eval_fn(cfg, model)` defaults to `PTrainer.test` over cfg.DATASETS.TEST), PeriodicWriter every 20 iterations
        (console line + one JSON record per line in OUTPUT_DIR/metrics.json, D2's JSONWriter format; evaluation results ride
        on the next record, as D2's storage does)."""
        import json
        import os
        from .. import checkpoint
        if start_iter is not None:
            self.iter = self.start_iter = start_iter
        max_iter = max_iter or self.max_iter
        rank0 = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
        ckpt = checkpoint.PeriodicCheckpointer(self, self.cfg.SOLVER.CHECKPOINT_PERIOD, max_iter) if rank0 else None
        out_dir = self.cfg.OUTPUT_DIR
        if rank0:
            os.makedirs(out_dir, exist_ok=True)
        t_last, it_last = time.perf_counter(), self.iter
        period = int(self.cfg.TEST.EVAL_PERIOD)
        eval_fn = eval_fn or type(self).test
        pending_eval: Dict[str, float] = {}
        while self.iter < max_iter:
            it = self.iter
            m = self.run_step()
            if ckpt is not None:
                ckpt.step(it)
            # D2 EvalHook: after every `period`-th iteration (after_step) and once after the last one (after_train)
            if run_eval and ((period > 0 and (it + 1) % period == 0) or it + 1 >= max_iter):
                flat = self._run_eval_hooks(eval_fn)
                pending_eval.update(flat)
                if rank0 and flat:
                    print(f"eval @ iter {it}: " + "  ".join(f"{k}: {v:.4f}" for k, v in sorted(flat.items())), flush=True)
            if rank0 and ((it + 1) % log_period == 0 or it == max_iter - 1):
                now = time.perf_counter()
                rec = dict(m, iteration=it, lr=lr_at(self.cfg, it), time=(now - t_last) / max(it + 1 - it_last, 1))
                rec.update(pending_eval)
                pending_eval = {}
                t_last, it_last = now, it + 1
                with open(os.path.join(out_dir, "metrics.json"), "a") as f:
                    f.write(json.dumps(rec, sort_keys=True) + "\n")
                print(f"iter: {it}  total_loss: {m['total_loss']:.4f}  " +
                      "  ".join(f"{k}: {v:.4f}" for k, v in m.items() if k[:4] == "loss") +
                      f"  time: {rec['time']:.4f}  lr: {rec['lr']:.6f}", flush=True)
        return self.last_metrics


def _flatten_like(model, ref: FlatParams) -> FlatParams:
    """Flatten `model` with exactly the layout of `ref` (so student/teacher buffers line up element for element)."""
    named = dict(model.named_parameters())
    fp = FlatParams.__new__(FlatParams)
    fp.flat = torch.empty_like(ref.flat)
    fp.index = ref.index
    fp.n_trainable = ref.n_trainable
    fp.params = {}
    fp.grad = None
    fp.on_zero_grad = []
    for n, (off, k) in ref.index.items():
        p = named[n]
        view = fp.flat[off:off + k].view(p.shape)
        view.copy_(p.data)
        p.data = view
        fp.params[n] = p
    return fp
