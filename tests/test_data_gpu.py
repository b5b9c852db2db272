"""GPU (pytest -m gpu): the non-synthetic entry path end to end -- VOC-format files on disk -> host decode -> device two-crop
mapper -> aspect-ratio grouping -> PTrainer(data_loader=...) -> train() with the student / teacher eval hooks on
cfg.DATASETS.TEST (reference train_net.py:51-75, pt/data/build.py:107-217, pt/engine/trainer.py:498-547)."""
import json
import os

import numpy as np
import pytest
import torch

from tests.test_host_logic import _write_voc_dir

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_three_iterations_from_files_with_eval_hooks(tmp_path):
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.data import datasets
    from probabilisticteacher_amd.engine import PTrainer
    rng = np.random.RandomState(4)
    names = ("car",)
    for sub, n in (("label", 6), ("unlabel", 6), ("val", 3)):
        _write_voc_dir(str(tmp_path / sub), [f"{sub}{i}" for i in range(n)], names, rng, h=160, w=224)
        os.rename(tmp_path / sub / "ImageSets" / "Main" / "train.txt", tmp_path / sub / "ImageSets" / "Main" / "split.txt")
        datasets.register_pascal_voc("t_" + sub, str(tmp_path / sub), "split", names)
    out = tmp_path / "out"
    cfg = setup_cfg("configs/pt/final_s2c.yaml", [
        "MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 1, "SOLVER.IMG_PER_BATCH_LABEL", 2,
        "SOLVER.IMG_PER_BATCH_UNLABEL", 2, "DATASETS.TRAIN_LABEL", ("t_label",), "DATASETS.TRAIN_UNLABEL", ("t_unlabel",),
        "DATASETS.TEST", ("t_val",), "TEST.EVAL_PERIOD", 2, "INPUT.MIN_SIZE_TRAIN", (160,), "INPUT.MAX_SIZE_TRAIN", 320,
        "INPUT.MIN_SIZE_TEST", 160, "INPUT.MAX_SIZE_TEST", 320, "OUTPUT_DIR", str(out), "SOLVER.CHECKPOINT_PERIOD", 100])
    torch.manual_seed(0)
    loader = PTrainer.build_train_loader(cfg)
    first = next(loader)
    assert len(first) == 4 and all(len(s) == 2 for s in first), "(label_strong, label_weak, unlabel_strong, unlabel_weak) x batch 2"
    ls, lw, us, uw = first
    assert ls[0]["image"].is_cuda and ls[0]["image"].dtype == torch.uint8 and ls[0]["image"].shape[0] == 3
    assert min(ls[0]["image"].shape[-2:]) == 160, "ResizeShortestEdge(MIN_SIZE_TRAIN) on the device"
    assert ls[0]["instances"].has("gt_boxes") and len(ls[0]["instances"].gt_boxes) >= 1 and "instances" not in us[0]
    assert not torch.equal(ls[0]["image"], lw[0]["image"]), "strong view != weak view"
    tr = PTrainer(cfg, data_loader=PTrainer.build_train_loader(cfg))
    m = tr.train(max_iter=3, log_period=1)                       # burn-in, EMA copy + mutual learning x 2; eval after 2 and 3
    assert np.isfinite(m["total_loss"]) or any(np.isnan(v) for k, v in m.items() if k.endswith("_unsup"))
    recs = [json.loads(line) for line in open(out / "metrics.json")]
    assert [r["iteration"] for r in recs] == [0, 1, 2]
    with_eval = [r for r in recs if "bbox/AP50" in r]
    assert len(with_eval) == 2 and all("bbox_student/AP50" in r and 0.0 <= r["bbox/AP50"] <= 100.0 for r in with_eval)
    assert tr._last_eval_results_student["bbox"].keys() == {"AP", "AP50", "AP75"}
    # --eval-only path: DefaultTrainer.test over cfg.DATASETS.TEST
    res = PTrainer.test(cfg, tr.model)
    assert set(res["bbox"]) == {"AP", "AP50", "AP75"} and tr.model.training
