#!/usr/bin/env python
"""train_net.py -- CLI equivalent of the reference's train_net.py / train.sh for the MI355X-native step.

    python train_net.py --num-gpus 1 --config configs/pt/final_c2f.yaml \
        MODEL.ANCHOR_GENERATOR.NAME DifferentiableAnchorGenerator UNSUPNET.TAU [0.5,0.5] MODEL.VGG.PRETRAIN ''
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_net.py --config ... --synthetic

Same flags as detectron2's default_argument_parser that the reference relies on (--config-file with argparse prefix
matching, --num-gpus, opts); process launch is torchrun-style (one process per GPU, RCCL).

Data: by default the datasets named by cfg.DATASETS.TRAIN_LABEL / TRAIN_UNLABEL / TEST (VOC-format directories registered
under $DETECTRON2_DATASETS as in pt/data/datasets/builtin.py, or ad hoc with --register NAME=DIR:SPLIT:CLASSES) go through
the host decoder + device two-crop mapper + aspect-ratio grouping (probabilisticteacher_amd/data, reference
pt/data/build.py:107-217); `--synthetic` feeds seeded synthetic two-crop batches in the same record format instead."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def synthetic_loader(cfg, dev, rank, world):
    from bench import synth_records
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    bl, bu = cfg.SOLVER.IMG_PER_BATCH_LABEL // world, cfg.SOLVER.IMG_PER_BATCH_UNLABEL // world
    assert bl >= 1 and bu >= 1, "batch must be divisible by the world size (pt/data/build.py:174-187)"
    gen = torch.Generator().manual_seed(1234 + rank * 1000)
    h, w = 800, 1333
    while True:
        yield (synth_records(gen, bl, h, w, K, dev), synth_records(gen, bl, h, w, K, dev),
               synth_records(gen, bu, h, w, K, dev), synth_records(gen, bu, h, w, K, dev))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-file", default="", metavar="FILE")
    ap.add_argument("--num-gpus", type=int, default=1)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--max-iter", type=int, default=None)
    ap.add_argument("--resume", action="store_true",
                    help="continue from MODEL.WEIGHTS (or OUTPUT_DIR/last_checkpoint): weights, optimiser state, iteration")
    ap.add_argument("--eval-only", action="store_true")
    ap.add_argument("--register", action="append", default=[], metavar="NAME=DIR:SPLIT:CLS1,CLS2",
                    help="register a VOC-format directory as dataset NAME (Annotations/, JPEGImages/, ImageSets/Main/SPLIT.txt)")
    ap.add_argument("opts", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer
    cfg = setup_cfg(args.config_file, ["MODEL.DEVICE", f"cuda:{local}"] + args.opts)
    if cfg.UNSUPNET.Trainer != "pt":
        raise ValueError("Trainer Name is not found.")
    from probabilisticteacher_amd.data import datasets
    for spec in args.register:
        name, rest = spec.split("=", 1)
        dirname, split, classes = rest.split(":")
        datasets.register_pascal_voc(name, dirname, split, tuple(classes.split(",")))
    torch.manual_seed(0)
    if args.synthetic:
        loader = synthetic_loader(cfg, torch.device("cuda", local), rank, world)
    else:
        loader = PTrainer.build_train_loader(cfg)              # trainer.py:139-141 -> pt/data/build.py:107
    trainer = PTrainer(cfg, data_loader=loader)
    inc = trainer.resume_or_load(resume=args.resume)          # trainer.py:466-496 (no-op without MODEL.WEIGHTS / checkpoint)
    if inc is not None and rank == 0:
        print(f"loaded {cfg.MODEL.WEIGHTS or 'last_checkpoint'}: start_iter {trainer.start_iter}, "
              f"missing {len(inc.missing_keys)}, unexpected {len(inc.unexpected_keys)}, wrong shape {len(inc.incorrect_shapes)}",
              flush=True)
    if args.eval_only and not args.synthetic:
        # reference train_net.py:60-75: evaluate the STUDENT of the loaded ensemble on cfg.DATASETS.TEST
        res = PTrainer.test(cfg, trainer.model)
        if rank == 0:
            print(res, flush=True)
        return
    if args.eval_only:
        n_batches = args.max_iter or 4
        loader = (next(trainer._data_iter)[1] for _ in range(n_batches))         # the weak labelled views
        names = [f"class{i}" for i in range(cfg.MODEL.ROI_HEADS.NUM_CLASSES)]
        ecfg = cfg.clone()
        ecfg.defrost()
        ecfg.TEST.EVALUATOR = "VOCeval"
        res = PTrainer.test(ecfg, trainer.model, loader, names)
        if rank == 0:
            print({k: v for k, v in res.items() if k == "bbox"}, flush=True)
        return
    # periodic checkpoints, metrics.json, model_final.pth; the student / teacher eval hooks run on cfg.DATASETS.TEST
    trainer.train(max_iter=args.max_iter, run_eval=not args.synthetic)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
