#!/usr/bin/env python
"""bench.py -- teacher+student train-step throughput (img/s) at 1333x800 on N x MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full PTrainer.run_step in the mutual-learning phase (BASELINE.json configs[2]): EMA teacher update,
teacher forward on B unlabelled weak views, student supervised forward on 2B labelled views, student unsupervised
forward on B strong views, ONE backward, gradient all-reduce (RCCL) for N>1, clip + SGD.  Per GPU B_label =
B_unlabel = --per-gpu-batch (weak scaling: per-GPU work fixed).  img/s = N*(B_label+B_unlabel)/step_seconds
(SURVEY.md 8d).  Synthetic uint8 images resident in HBM before the timed region, random-init weights, fp32.

Prints ONE JSON line (rank 0) with `roofline` (MFMA conv kernel, HIP-event timed inside the timed region) and,
at N=1, `cpu_baseline` (the CPU oracle on the host cores, bounded sample)."""
import argparse
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: v_mfma_f32_32x32x16_bf16 dense peak (--amp only)


def synth_records(gen, n, h, w, K, dev, m=12, labelled=True):
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    recs = []
    for _ in range(n):
        img = torch.randint(0, 256, (3, h, w), generator=gen, dtype=torch.uint8).to(dev)
        r = {"image": img, "height": h, "width": w}
        if labelled:
            bw = torch.exp(torch.rand(m, generator=gen) * (math.log(400) - math.log(32)) + math.log(32))
            bh = torch.exp(torch.rand(m, generator=gen) * (math.log(400) - math.log(32)) + math.log(32))
            cx, cy = torch.rand(m, generator=gen) * w, torch.rand(m, generator=gen) * h
            b = torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
            b[:, 0::2].clamp_(0, w)
            b[:, 1::2].clamp_(0, h)
            keep = ((b[:, 2] - b[:, 0]) > 4) & ((b[:, 3] - b[:, 1]) > 4)
            inst = FreeInstances((h, w))
            inst.gt_boxes = Boxes(b[keep].to(dev))
            inst.gt_classes = torch.randint(0, K, (int(keep.sum()),), generator=gen).to(dev)
            r["instances"] = inst
        recs.append(r)
    return recs


def cpu_baseline(h, w, K, seed=0, student_only=False):
    """The CPU oracle (oracle/pt.py, a port of the reference's algorithm on stock torch CPU fp32 ops) timed on this
    box's host cores: ONE full mutual-learning step with 1 labelled + 1 unlabelled image (bounded sample)."""
    from oracle import d2, pt as opt
    torch.set_num_threads(min(os.cpu_count() or 1, 32))      # oneDNN scales poorly past ~32 threads at batch 1
    cfg = opt.Cfg(num_classes=K, burn_up_step=10 ** 9 if student_only else 0)
    gen = torch.Generator().manual_seed(seed)
    state = {"student": opt.init_params(cfg, 0), "teacher": opt.init_params(cfg, 0), "bufs": {}, "iter": 0}

    def recs(labelled=True):
        r = {"image": torch.randint(0, 256, (3, h, w), generator=gen, dtype=torch.uint8), "height": h, "width": w}
        inst = opt.FreeInstances((h, w))
        b = torch.tensor([[100.0, 120.0, 400.0, 380.0], [600.0, 200.0, 900.0, 700.0], [50.0, 500.0, 300.0, 780.0]])
        inst.gt_boxes = d2.Boxes(b)
        inst.gt_classes = torch.tensor([0, 1, 2]) % K
        r["instances"] = inst
        return [r]
    data = (recs(), recs(), recs(), recs())
    t0 = time.perf_counter()
    opt.run_step(cfg, state, data, {"label": [0.8, 0.75], "unlabel": [0.7]}, perm_fn=opt.SeededPerm(1))
    dt = time.perf_counter() - t0
    return {"value": 2.0 / dt, "unit": "img/s", "cores": torch.get_num_threads(), "host_cores_total": os.cpu_count(),
            "kind": "port",
            "sample": (f"1 supervised student step (2 fwd, 2 bwd, clip+SGD) on the strong + weak view of 1 labelled {w}x{h} "
                       f"image, {dt:.1f} s" if student_only else
                       f"1 full teacher+student step (EMA, teacher fwd, 3 student fwd, 3 bwd, clip+SGD) with 1 labelled + "
                       f"1 unlabelled {w}x{h} image, {dt:.1f} s")}


PMC_PROFILE = "profiles/r06_bench_b16_pmc_by_kernel.json"
DOMINANT = "conv3x3_wino4p_kernel"        # the kernel the roofline object describes (its rocprofv3 name contains this)
DOMINANT_AMP = "p8_conv3x3_kernel"        # ... with --amp: the bf16-storage forward / dgrad kernel


def committed_pmc_traffic():
    """Fallback only: HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (tools/pmc_collect.py).  Returns (bytes per launch, provenance) or (None, reason)."""
    f = os.path.join(ROOT, PMC_PROFILE)
    if not os.path.exists(f):
        return None, f"{PMC_PROFILE} absent"
    d = json.load(open(f))
    tot, n = 0.0, 0
    for k, v in d.get("kernels", {}).items():
        if DOMINANT in k:
            tot += (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0
            n += v["dispatches"]
    return (tot / n if n else None), f"{PMC_PROFILE} (collected on {d.get('git_head', '?')}, {d.get('command', '?')})"


def measured_pmc_traffic(args):
    """HBM-side bytes per launch of the dominant kernel, measured NOW: this script re-runs itself for one warm-up + one step
    under `rocprofv3 --kernel-trace --pmc <counter>` -- FETCH_SIZE and WRITE_SIZE in separate passes, as
    MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 64 B per 128-B request on gfx950: doubled) -- after the timed region,
    so that the counters do not perturb the timing.  Returns (bytes per launch, provenance) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    tot, disp = {}, 0
    dominant = DOMINANT_AMP if args.amp else DOMINANT
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ptmi_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--output-format", "csv", "--",
               sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
               "--pmc-traffic", "off", "--per-gpu-batch", str(args.per_gpu_batch), "--height", str(args.height),
               "--width", str(args.width)] + (["--student-only"] if args.student_only else []) + (["--amp"] if args.amp else [])  # (--config3 is already folded into --per-gpu-batch)
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=420)
        except Exception as e:      # noqa: BLE001  (a profiler failure must not take the bench line with it)
            shutil.rmtree(d, ignore_errors=True)
            return None, f"rocprofv3 {counter} pass failed: {e!r}"
        n, val = 0, 0.0
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if dominant in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    val += float(row["Counter_Value"])
                    n += 1
        shutil.rmtree(d, ignore_errors=True)
        if r.returncode != 0 or n == 0:
            return None, f"rocprofv3 {counter} pass: rc {r.returncode}, {n} dispatches of {dominant}: {r.stdout[-300:]!r}"
        tot[counter], disp = val, n
    return ((2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / disp,
            f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, 1 warm-up + 1 step each, "
            f"{disp} dispatches of {dominant}; FETCH_SIZE x2 per the gfx950 correction)")


def extra_leg(flags, steps=20, warmup=5, pmc="auto"):
    """One more leg of the SAME script in a fresh process, after the headline's timed region (so that nothing of it perturbs the
    fp32 number): returns the leg's own JSON line reduced to what the headline line carries for it, or {"error": ...}."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup),
           "--no-cpu-baseline", "--no-extras", "--pmc-traffic", pmc] + flags
    try:
        r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
    except Exception as e:      # noqa: BLE001  (an extra leg must not take the headline line with it)
        return {"error": repr(e)[:300]}
    rf = d["roofline"]
    return {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
            "ms_per_step_median": d["ms_per_step_median"], "steps": d["steps"], "warmup": d["warmup"], "dtype": d["dtype"],
            "workload": d["config"]["workload"], "global_batch": d["config"]["global_batch"],
            "roofline": {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                            "algorithmic_bytes_per_launch", "kernel", "avg_launch_ms", "calls",
                                            "effective_direct_tflops")},
            "kernels": d["kernels"], "losses": d["losses"], "command": " ".join(["python", "bench.py"] + cmd[2:])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--per-gpu-batch", type=int, default=16, help="B_label = B_unlabel per GPU")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--student-only", action="store_true",
                    help="BASELINE configs[1]: supervised student fwd/bwd only (burn-in step) on 2 x per-gpu-batch images; "
                         "use --per-gpu-batch 4 for the quoted batch of 8")
    ap.add_argument("--amp", action="store_true",
                    help="NOT the headline metric: SOLVER.AMP.ENABLED (BASELINE configs[4] numerics) -- bf16 activations and "
                         "activation gradients in HBM / LDS for the 3x3 conv stack (ptmi_p8_*, v_mfma_f32_32x32x16_bf16), bf16 "
                         "operands for the FC GEMMs, fp32 accumulation, fp32 losses / optimiser; reported against the dense bf16 "
                         "MFMA peak")
    ap.add_argument("--config3", action="store_true",
                    help="BASELINE configs[3]: global batch 64 labelled + 64 unlabelled on 8 GPUs = 8 + 8 per GPU (SURVEY 8d C4; "
                         "pt/data/build.py:174-187 gives every rank total / world); sets --per-gpu-batch 8, so `--gpus N --config3` "
                         "for N = 1, 2, 4, 8 is that config's weak-scaling curve and N = 8 its own number")
    ap.add_argument("--grad-reduce", default="all_reduce", choices=["all_reduce", "reduce_scatter"],
                    help="N > 1: the bucketed gradient exchange as all-reduce or as reduce-scatter + all-gather (engine/flat.py)")
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1 default run only: skip the two extra legs that the JSON line carries next to the fp32 headline -- "
                         "`amp` (the same workload with SOLVER.AMP.ENABLED, BASELINE configs[4]'s precision) and `student_only` "
                         "(BASELINE configs[1], batch 8)")
    ap.add_argument("--tile-schedule", default=None, choices=["static", "dynamic"],
                    help="tile schedule of the persistent F(4x4,3x3) kernel (default: PTrainer's choice -- static on one GPU, dynamic "
                         "work queues when the gradient exchange is active; csrc/wino4.hip)")
    ap.add_argument("--batches", type=int, default=2,
                    help="distinct synthetic batches resident in HBM, rotated over the steps (the `sustained` leg uses 4)")
    ap.add_argument("--pmc-traffic", default="auto", choices=["auto", "off"],
                    help="auto (N = 1): measure roofline.traffic after the timed region by re-running one step under rocprofv3 "
                         "PMC passes; off: report the committed profile's number, labelled as such")
    args = ap.parse_args()
    if args.config3:
        args.per_gpu_batch = 8

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP extension is the only compute path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)          # RCCL over xGMI
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"

    from probabilisticteacher_amd import ops
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer

    B = args.per_gpu_batch
    cfg = setup_cfg(os.path.join(ROOT, "configs/pt/final_c2f.yaml"), [
        "MODEL.DEVICE", f"cuda:{local_rank}", "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP",
        10 ** 9 if args.student_only else 0,
        "SOLVER.IMG_PER_BATCH_LABEL", B * world, "SOLVER.IMG_PER_BATCH_UNLABEL", B * world,
        "SOLVER.AMP.ENABLED", bool(args.amp)])
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    torch.manual_seed(0)                                                # identical init on all ranks
    trainer = PTrainer(cfg, grad_reduce=args.grad_reduce)
    if args.tile_schedule:
        ops.set_tile_schedule(args.tile_schedule)
    gen = torch.Generator().manual_seed(1234 + rank * 1000)
    H, W = args.height, args.width

    def batch():
        return (synth_records(gen, B, H, W, K, dev), synth_records(gen, B, H, W, K, dev),
                synth_records(gen, B, H, W, K, dev), synth_records(gen, B, H, W, K, dev))
    NB = max(1, args.batches)
    batches = [batch() for _ in range(NB)]                              # resident in HBM before timing

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # memory-pool warm-up: one big segment for the caching allocator, so that the steps split cached blocks instead of
    # calling hipMalloc (the pool otherwise keeps growing for a few steps as ROI counts vary)
    total_mem = torch.cuda.get_device_properties(dev).total_memory
    pool = torch.empty(min(64 << 30, total_mem // 4), dtype=torch.uint8, device=dev)
    del pool
    for i in range(args.warmup):
        trainer.run_step(batches[i % NB])
    sync()
    ops.profile_start()
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    step_ms, early = [], []
    for i in range(args.steps):
        ts = time.perf_counter()
        trainer.run_step(batches[i % NB])         # (ends with the metrics read-back, i.e. synchronised)
        step_ms.append(1e3 * (time.perf_counter() - ts))
        early.append(trainer.reducer.launched_in_backward)
    sync()
    dt = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats()
    if rank == 0:
        print(f"[bench] per-step ms {[round(v, 1) for v in step_ms]}; device mallocs in the timed region: "
              f"{ms1.get('num_device_alloc', 0) - ms0.get('num_device_alloc', 0)}, reserved "
              f"{ms1.get('reserved_bytes.all.current', 0) / 2**30:.1f} GiB", file=sys.stderr)
    prof = ops.profile_stop()
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    # self-verification of the data-parallel run (all ranks take part): what the process group really looks like, every
    # rank's own step time, and that the replicas are still bit-identical after the timed steps
    from probabilisticteacher_amd.engine.flat import replicas_identical
    per_rank = torch.tensor([dt / args.steps * 1e3], dtype=torch.float64, device=dev)
    if world > 1:
        gathered = [torch.empty_like(per_rank) for _ in range(world)]
        dist.all_gather(gathered, per_rank)
        per_rank_ms = [float(g.item()) for g in gathered]
    else:
        per_rank_ms = [float(per_rank.item())]
    same_s, sums_s = replicas_identical(trainer.student.flat)
    same_t, _ = replicas_identical(trainer.teacher.flat)
    assert same_s and same_t, f"data-parallel replicas diverged: student checksums {sums_s}"
    dt = dt_max

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * 2 * B * args.steps / dt             # burn-in step: label_q + label_k = 2B images as well
        srt = sorted(step_ms)
        median = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
        traffic, traffic_src = (measured_pmc_traffic(args) if args.pmc_traffic == "auto" and world == 1
                                else (None, "not measured in this run (--pmc-traffic off or N > 1)"))
        if traffic is None and not args.amp:
            fb, fb_src = committed_pmc_traffic()
            traffic, traffic_src = fb, f"{fb_src}; {traffic_src}"
        peak = PEAK_BF16_MFMA_TFLOPS if args.amp else PEAK_F32_MFMA_TFLOPS
        # the dominant kernel: the fused Winograd F(2x2,3x3) kernel (fp32 bench) / the bf16-storage direct kernel (--amp).
        # `achieved` / `frac` price the MFMA FLOPs the kernel ISSUES (never above the peak); the direct-convolution FLOPs the
        # same launches stand for are reported next to it as `effective_direct_tflops`
        dom = "p8_conv3x3" if args.amp else "conv3x3_wino4"
        conv = prof.get(dom, {"ms": 0.0, "flops": 0.0, "issued": 0.0, "bytes": 0.0, "calls": 0})
        ach = conv["issued"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
        eff = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
        out = {
            "metric": ("student-only train-step img/s at 1333x800" if args.student_only else
                       "teacher-student train-step img/s at 1333x800") +
                      (" [SOLVER.AMP.ENABLED: bf16 operands, not the headline metric]" if args.amp else ""),
            "value": value, "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "ms_per_step_median": median, "value_at_median": world * 2 * B / (median * 1e-3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 storage + operands / f32 accumulate (3x3 conv stack), bf16 operands (FC); f32 elsewhere" if args.amp else "f32",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[1]: final_c2f.yaml (K=8) student-only supervised fwd/bwd + clip + SGD, "
                                    f"per-GPU {2 * B} synthetic {W}x{H} images (strong + weak view of {B} labelled), random init"
                                    if args.student_only else
                                    f"BASELINE configs[{'3' if args.config3 else '2'}]: final_c2f.yaml (K=8) full teacher+student+EMA step, per-GPU "
                                    f"{B} labelled + {B} unlabelled synthetic {W}x{H} images, BURN_UP_STEP=0, random init"),
                       "global_batch": 2 * B * world, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                         "frac": ach / peak, "traffic": traffic,
                         "traffic_unit": "HBM-side bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": conv["bytes"] / max(conv["calls"], 1),
                         "kernel": ("p8_conv3x3_kernel<MT> (bf16 storage: all 3x3 conv fwd + dgrad launches)" if args.amp else
                                    "conv3x3_wino4p_kernel (fused Winograd F(4x4,3x3), positions split over the wave pair, round 6: every 3x3 conv fwd + dgrad launch with "
                                    ">= 64 input channels; the 3-channel stem runs conv3x3_stem_kernel, listed under kernels)"),
                         "achieved_is": "MFMA FLOPs issued by the kernel (36 multiplies per 4x4 tile and channel pair -- a quarter of "
                                        "the direct algorithm's 144 -- tile padding included) / HIP-event time; equals the direct "
                                        "FLOPs for the bf16 kernels.  effective_direct_tflops prices the same launches by the direct "
                                        "convolution's FLOPs (SURVEY 8d's per-image figure)",
                         "effective_direct_tflops": eff,
                         "effective_direct_over_peak": eff / peak,
                         "calls": conv["calls"],
                         "avg_launch_ms": conv["ms"] / max(conv["calls"], 1),
                         "mfma_flops_issued_per_launch_avg": conv["issued"] / max(conv["calls"], 1),
                         "direct_flops_per_launch_avg": conv["flops"] / max(conv["calls"], 1)},
            "kernels": {k: {"ms_per_step": v["ms"] / args.steps, "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12)
                            if v["ms"] > 0 and v["flops"] else None, "calls_per_step": v["calls"] / args.steps}
                        for k, v in prof.items()},
            "step_conv_tflops": ((0.67106 + 0.90253) if args.student_only else 2.696) * 2 * B * world * args.steps / dt
            if (H, W) == (800, 1333) else None,
            "losses": {k: v for k, v in trainer.last_metrics.items() if k.startswith("loss")},
            "distributed": {"world_size_seen_by_process_group": dist.get_world_size() if dist.is_initialized() else 1,
                            "backend": dist.get_backend() if dist.is_initialized() else None,
                            "per_rank_ms_per_step": per_rank_ms,
                            "grad_exchange": {"mode": trainer.reducer.mode, "active": trainer.reducer.active,
                                              "buckets": len(trainer.reducer.buckets),
                                              "buckets_launched_during_backward_avg": sum(early) / max(len(early), 1),
                                              "overlap_fraction": (sum(early) / max(len(early), 1)) / len(trainer.reducer.buckets),
                                              "bytes_exchanged_per_step": trainer.reducer.bytes_per_step if trainer.reducer.active else 0},
                            "replicas_bit_identical_after_run": bool(same_s and same_t),
                            "student_checksums": [str(v) for v in sums_s]},
        }
        if world == 1 and not args.no_cpu_baseline and not args.amp:
            out["cpu_baseline"] = cpu_baseline(H, W, K, student_only=args.student_only)
        if world == 1 and not args.no_extras and not args.amp and not args.student_only and not args.config3:
            # BASELINE configs[4]'s precision and configs[1] on the driver's record: the same PTrainer workload with
            # SOLVER.AMP.ENABLED (pt/engine/trainer.py:98), and the student-only burn-in step at batch 8 -- each in its own
            # process after the fp32 timed region; the headline keys above stay fp32
            del trainer, batches
            torch.cuda.empty_cache()
            size = ["--height", str(H), "--width", str(W)]
            out["amp"] = extra_leg(["--amp", "--per-gpu-batch", str(B)] + size)
            out["student_only"] = extra_leg(["--student-only", "--per-gpu-batch", "4"] + size, pmc="off")
            # the headline workload again, 100 timed steps over FOUR rotating batches (ROI counts vary with the batch, so the
            # allocator and the variable-size kernels see more shapes than in the 20-step / 2-batch headline)
            out["sustained"] = extra_leg(["--per-gpu-batch", str(B), "--batches", "4"] + size, steps=100, pmc="off")

            def compact(leg, with_roof=True):
                if "error" in leg:
                    return {"error": leg["error"][:60]}
                c = {"value": round(leg["value"], 2), "ms_per_step": round(leg["ms_per_step"], 2)}
                if with_roof:
                    rf = leg["roofline"]
                    c["frac"] = round(rf["frac"], 4)
                    measured = str(rf.get("traffic_source", "")).startswith("measured in this run")   # (not the committed-profile fallback)
                    c["traffic_ratio"] = (round(rf["traffic"] / rf["algorithmic_bytes_per_launch"], 3)
                                          if measured and rf.get("traffic") and rf.get("algorithmic_bytes_per_launch") else None)
                return c
            # LAST key of the line, compact: what survives in a record that keeps only the tail of stdout
            legs = {"amp": compact(out["amp"]), "student_only": compact(out["student_only"]),
                    "sustained": dict(compact(out["sustained"], False), steps=100, batches=4)}
            out["legs"] = legs
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
