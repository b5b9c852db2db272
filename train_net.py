#!/usr/bin/env python
"""train_net.py -- CLI equivalent of the reference's train_net.py / train.sh for the MI355X-native step.

    python train_net.py --num-gpus 1 --config configs/pt/final_c2f.yaml \
        MODEL.ANCHOR_GENERATOR.NAME DifferentiableAnchorGenerator UNSUPNET.TAU [0.5,0.5] MODEL.VGG.PRETRAIN ''
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_net.py --config ... --synthetic

Same flags as detectron2's default_argument_parser that the reference relies on (--config-file with argparse prefix
matching, --num-gpus, opts); process launch is torchrun-style (one process per GPU, RCCL).  The reference's data
pipeline (pt/data) is out of scope for this build: `--synthetic` feeds seeded synthetic
two-crop batches in the reference's record format; a real loader can be passed to PTrainer(data_loader=...)."""
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
    if not args.synthetic:
        raise SystemExit("only --synthetic input is available in this build (the data pipeline is out of scope)")
    torch.manual_seed(0)
    trainer = PTrainer(cfg, data_loader=synthetic_loader(cfg, torch.device("cuda", local), rank, world))
    inc = trainer.resume_or_load(resume=args.resume)          # trainer.py:466-496 (no-op without MODEL.WEIGHTS / checkpoint)
    if inc is not None and rank == 0:
        print(f"loaded {cfg.MODEL.WEIGHTS or 'last_checkpoint'}: start_iter {trainer.start_iter}, "
              f"missing {len(inc.missing_keys)}, unexpected {len(inc.unexpected_keys)}, wrong shape {len(inc.incorrect_shapes)}",
              flush=True)
    if args.eval_only:
        # reference train_net.py:60-75: evaluate the STUDENT of the loaded ensemble; here on synthetic labelled batches
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
    trainer.train(max_iter=args.max_iter)                      # periodic checkpoints, metrics.json, model_final.pth
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
