#!/usr/bin/env python
"""CU contention on ONE GPU: the 8 + 8 step of BASELINE configs[3] (its per-GPU share) while a side
stream holds k CUs for the whole step -- what the kernels of an overlapped RCCL collective do to the compute stream under DDP
(reference pt/engine/trainer.py:92-95,384).  ptmi_hold_cus: k workgroups with 96 KB of LDS each, so no persistent convolution
workgroup (156 KB) fits beside one.

    python tools/exp/contention.py [--hold 0,8,16,32] [--schedule static,dynamic] [--lib <variant .so>] [--steps 4]

Prints, per (schedule, k): ms per step, the ratio to k = 0 and to the "fair share" 256 / (256 - k), and the per-kernel-group times
of the groups that moved."""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hold", default="0,8,16,32")
    ap.add_argument("--schedule", default="static,dynamic")
    ap.add_argument("--lib", default=None)
    ap.add_argument("--waves", default="1", help="comma list: weight-gradient waves of workgroups (ops.set_wgrad_waves)")
    ap.add_argument("--amp", action="store_true", help="SOLVER.AMP.ENABLED: the bf16-storage kernels (--p8-waves: their convolution's fills)")
    ap.add_argument("--p8-waves", default="1", help="comma list, paired with --waves by position (ops.set_p8_conv_waves)")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--per-gpu-batch", type=int, default=8)
    args = ap.parse_args()
    from probabilisticteacher_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from bench import synth_records
    from probabilisticteacher_amd import ops
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer
    dev = torch.device("cuda", 0)
    B = args.per_gpu_batch
    cfg = setup_cfg(os.path.join(ROOT, "configs/pt/final_c2f.yaml"), [
        "MODEL.DEVICE", "cuda:0", "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 0,
        "SOLVER.IMG_PER_BATCH_LABEL", B, "SOLVER.IMG_PER_BATCH_UNLABEL", B, "SOLVER.AMP.ENABLED", bool(args.amp)])
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    torch.manual_seed(0)
    trainer = PTrainer(cfg)
    gen = torch.Generator().manual_seed(1234)
    batches = [tuple(synth_records(gen, B, 800, 1333, K, dev) for _ in range(4)) for _ in range(2)]
    pool = torch.empty(min(48 << 30, torch.cuda.get_device_properties(dev).total_memory // 4), dtype=torch.uint8, device=dev)
    del pool
    side = torch.cuda.Stream()
    scratch = torch.zeros(1, dtype=torch.int32, device=dev)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    print(f"# {cus} CUs; lib {_lib.LIB_PATH}; {B} + {B} images of 1333x800 per step")
    base = {}
    rows = []
    p8w = [int(v) for v in args.p8_waves.split(",")]
    for sched_w in [(s_, int(w_), p8w[min(i_, len(p8w) - 1)]) for s_ in args.schedule.split(",") for i_, w_ in enumerate(args.waves.split(","))]:
        sched = f"{sched_w[0]}/w{sched_w[1]}" + (f"/p{sched_w[2]}" if args.amp else "")
        ops.set_tile_schedule(sched_w[0])
        ops.set_wgrad_waves(sched_w[1])
        ops.set_p8_conv_waves(sched_w[2])
        for k in [int(v) for v in args.hold.split(",")]:
            hold_us = 0
            if k:
                hold_us = int(1e3 * base.get(sched, 250.0) * cus / (cus - k) * 1.6)
            ms = []
            prof = None
            for i in range(args.warmup + args.steps):
                torch.cuda.synchronize()
                if i == args.warmup:
                    ops.profile_start()
                if k:
                    with torch.cuda.stream(side):
                        _lib.call("ptmi_hold_cus", ops._ptr(scratch), k, hold_us, ctypes.c_void_p(side.cuda_stream))
                    time.sleep(0.002)                    # the holders are resident before the step's first launch
                t0 = time.perf_counter()
                trainer.run_step(batches[i % 2])         # (ends with the metrics read-back on the compute stream)
                torch.cuda.current_stream().synchronize()
                dt = 1e3 * (time.perf_counter() - t0)
                if i >= args.warmup:
                    ms.append(dt)
            prof = ops.profile_stop()
            torch.cuda.synchronize()
            m = sum(ms) / len(ms)
            if k == 0:
                base[sched] = m
            rows.append((sched, k, m, {n: v["ms"] / args.steps for n, v in prof.items()}))
            fair = cus / (cus - k)
            print(f"{sched:15s} hold {k:3d} CUs: {m:8.1f} ms/step  x{m / base[sched]:.3f} of hold 0 (fair share x{fair:.3f})  "
                  f"per-step {[round(v, 1) for v in ms]}")
    print("\n# per-kernel-group ms per step (groups >= 1 ms at hold 0)")
    names = [n for n, v in rows[0][3].items() if v >= 1.0]
    print("schedule hold " + " ".join(f"{n[:22]:>22s}" for n in names))
    for sched, k, m, pr in rows:
        print(f"{sched:15s} {k:4d} " + " ".join(f"{pr.get(n, 0.0):22.2f}" for n in names))
    print(json.dumps({"rows": [{"schedule": s, "hold": k, "ms": m, "kernels": pr} for s, k, m, pr in rows]}))


if __name__ == "__main__":
    main()
