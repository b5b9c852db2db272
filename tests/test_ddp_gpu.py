"""GPU (pytest -m gpu): the data-parallel gradient exchange on the ONE GPU a test box has.

`init_process_group("nccl", world_size=1)` (RCCL) + `PTrainer(force_grad_reducer=True)`: the bucketed all-reduces are
launched from the post-accumulate-grad hooks DURING the real backward of `run_step`, on RCCL's stream, against the flat
gradient buffer that autograd keeps accumulating into -- exactly the stream ordering an 8-GPU run has.  With one rank
the sum is the identity, so gradients, clipped update and parameters must equal the run without the reducer BITWISE."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture()
def rccl_world1():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    yield
    dist.destroy_process_group()


def _batch(gen, n, h, w, K):
    from bench import synth_records
    return tuple(synth_records(gen, n, h, w, K, DEV) for _ in range(4))


@pytest.mark.parametrize("mode", ["all_reduce", "reduce_scatter"])
def test_bucketed_rccl_allreduce_in_run_step_is_bitwise_identity(rccl_world1, mode):
    """`mode`: one all-reduce per bucket, or the same exchange as reduce-scatter + all-gather on shard-aligned buckets"""
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer
    from probabilisticteacher_amd.modeling import sampling
    cfg = setup_cfg("configs/pt/final_c2f.yaml", ["MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 1,
                                                  "SOLVER.IMG_PER_BATCH_LABEL", 2, "SOLVER.IMG_PER_BATCH_UNLABEL", 2])
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    results = []
    for force in (False, True):
        torch.manual_seed(0)
        ratios = iter([0.8, 0.6, 0.9, 0.7, 0.75, 0.65, 0.85, 0.55] * 2)
        tr = PTrainer(cfg, ratio_fn=lambda: next(ratios), force_grad_reducer=force, grad_reduce=mode)
        assert tr.reducer.active == force and len(tr.reducer.buckets) >= 4
        gen = torch.Generator().manual_seed(77)
        keyg = torch.Generator().manual_seed(5)
        sampling.set_key_source(lambda labels, sizes, bg: torch.rand(labels.shape, generator=keyg))
        try:
            out = []
            for it in range(2):                       # burn-in step (anchor table gets no gradient), then mutual learning
                m = tr.run_step(_batch(gen, 2, 320, 480, K))
                early = tr.reducer.launched_in_backward
                out.append((m, tr.student.grad.clone(), tr.student.flat.clone(), tr.teacher.flat.clone(), early))
        finally:
            sampling.set_key_source(None)
        results.append(out)
    for it, (a, b) in enumerate(zip(*results)):
        assert a[0].keys() == b[0].keys()
        for k in a[0]:
            if k != "data_time":
                assert a[0][k] == b[0][k], f"iteration {it}: metric {k}: {a[0][k]} vs {b[0][k]}"
        assert torch.equal(a[1], b[1]), f"iteration {it}: flat gradient must be bitwise equal"
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]), f"iteration {it}: parameters must be bitwise equal"
        # all buckets but the last one (which holds the first layers and the anchor table) left during backward
        assert a[4] == 0 and b[4] >= len(tr.reducer.buckets) - 1, (a[4], b[4], len(tr.reducer.buckets))


def test_metrics_all_gather_on_rccl(rccl_world1):
    """`_write_metrics` takes its multi-rank branch only for world_size > 1; drive the same collective on RCCL here"""
    v = torch.arange(26, dtype=torch.float32, device=DEV)
    out = torch.empty(26, device=DEV)
    dist.all_gather_into_tensor(out, v)
    assert torch.equal(out, v)


def test_checkpoint_resume_is_bitwise_on_gpu(tmp_path):
    """f3/f4 on the device: train 4 iterations straight through vs train 2, write a checkpoint (reference file format:
    EnsembleTSModel keys, per-parameter SGD state, iteration = last finished), build a FRESH trainer, resume_or_load(resume)
    and train 2 more -- parameters, teacher, momentum and metrics must be bitwise identical (burn-in -> EMA copy -> mutual
    learning boundary included: BURN_UP_STEP = 2 sits exactly at the resume point)."""
    from probabilisticteacher_amd import checkpoint
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer
    from probabilisticteacher_amd.modeling import sampling
    from bench import synth_records
    out = str(tmp_path)
    base = ["MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 2, "SOLVER.IMG_PER_BATCH_LABEL", 2,
            "SOLVER.IMG_PER_BATCH_UNLABEL", 2, "SOLVER.CHECKPOINT_PERIOD", 2, "OUTPUT_DIR", out]
    cfg = setup_cfg("configs/pt/final_c2f.yaml", base)
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    gen = torch.Generator().manual_seed(9)
    batches = [tuple(synth_records(gen, 2, 256, 384, K, DEV) for _ in range(4)) for _ in range(4)]

    def run(trainer, its, seed_base):
        ms = []
        for it in its:
            rr = iter([0.6 + 0.05 * j for j in range(8)])
            trainer._ratio_fn = lambda: next(rr)
            kg = torch.Generator().manual_seed(seed_base + it)
            sampling.set_key_source(lambda labels, sizes, bg: torch.rand(labels.shape, generator=kg))
            try:
                ms.append(trainer.run_step(batches[it]))
            finally:
                sampling.set_key_source(None)
        return ms

    torch.manual_seed(0)
    a = PTrainer(cfg)
    ma = run(a, range(4), 100)
    torch.manual_seed(0)
    b = PTrainer(cfg)
    mb = run(b, range(2), 100)
    path = checkpoint.PeriodicCheckpointer(b, cfg.SOLVER.CHECKPOINT_PERIOD, 4).step(1)
    assert path.endswith("model_0000001.pth")
    torch.manual_seed(123)                                    # a differently initialised trainer: everything must come from the file
    c = PTrainer(setup_cfg("configs/pt/final_c2f.yaml", base))
    assert not torch.equal(c.student.flat, b.student.flat)
    inc = c.resume_or_load(resume=True)                       # MODEL.WEIGHTS empty -> OUTPUT_DIR/last_checkpoint
    assert inc is not None and not inc.missing_keys and not inc.incorrect_shapes and c.iter == c.start_iter == 2
    assert torch.equal(c.student.flat, b.student.flat) and torch.equal(c.teacher.flat, b.teacher.flat)
    assert torch.equal(c.momentum_buf, b.momentum_buf) and not c._first_step
    mc = run(c, range(2, 4), 100)
    for x, y in zip(ma[2:], mc):
        for k in x:
            if k != "data_time":
                assert x[k] == y[k] or (x[k] != x[k] and y[k] != y[k]), f"metric {k}: {x[k]} vs {y[k]}"
    assert torch.equal(a.student.flat, c.student.flat) and torch.equal(a.teacher.flat, c.teacher.flat)
    assert torch.equal(a.momentum_buf, c.momentum_buf)
