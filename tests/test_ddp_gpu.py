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
    from probabilisticteacher_amd import ops
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
        # PTrainer picks the kernel policy by `reducer.active` (dynamic tile schedule + 3 weight-gradient waves under an exchange:
        # the split partial sums then associate differently); "the exchange is a bitwise identity" is a statement at EQUAL policy
        waves = PTrainer.ddp_wgrad_waves(False)
        assert waves == 3 and (ops._TILE_SCHEDULE, ops._WGRAD_WAVES) == (("dynamic", waves) if force else ("static", 1))
        ops.set_tile_schedule("dynamic")
        ops.set_wgrad_waves(waves)
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
            ops.set_tile_schedule("static")
            ops.set_wgrad_waves(1)
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


# ------------------------------------------------------------------------------------------------------------------------
# The REAL step on TWO ranks.  A test box has one GPU and RCCL refuses two ranks on one device, so the two processes share
# cuda:0 and talk over gloo.  gloo moves CUDA tensors for broadcast and all_reduce (the gradient exchange of the default
# mode, the start-up parameter broadcast); its all_gather / reduce_scatter take host tensors only, so those calls -- the
# metrics all-gather of `_write_metrics`, the checksum all-gather of `replicas_identical`, the reduce-scatter mode's two halves
# -- are bounced through host copies by the shims below (test-side only: the product code calls torch.distributed as it does
# on RCCL).
class _Done:
    def wait(self, *a, **k):
        return True


def _install_gloo_cuda_shims():
    o_agt, o_rst, o_ag = dist.all_gather_into_tensor, dist.reduce_scatter_tensor, dist.all_gather

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        if not inp.is_cuda:
            return o_agt(out, inp, group=group, async_op=async_op)
        host = torch.empty(out.shape, dtype=out.dtype)
        o_agt(host, inp.cpu(), group=group)
        out.copy_(host)
        return _Done() if async_op else None

    def reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not inp.is_cuda:
            return o_rst(out, inp, op=op, group=group, async_op=async_op)
        host = torch.empty(out.shape, dtype=out.dtype)
        o_rst(host, inp.cpu(), op=op, group=group)
        out.copy_(host)
        return _Done() if async_op else None

    def all_gather(outs, inp, group=None, async_op=False):
        if not inp.is_cuda:
            return o_ag(outs, inp, group=group, async_op=async_op)
        hosts = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        o_ag(hosts, inp.cpu(), group=group)
        for o, h in zip(outs, hosts):
            o.copy_(h)
        return _Done() if async_op else None

    dist.all_gather_into_tensor, dist.reduce_scatter_tensor, dist.all_gather = all_gather_into_tensor, reduce_scatter_tensor, all_gather


def _two_rank_worker(rank, world, port, mode, q, backend="gloo"):
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":                            # RCCL, one GPU per rank (needs >= 2 visible GPUs)
        DEV = f"cuda:{rank}"                         # noqa: N806  (shadows the module constant for everything below)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(DEV))
    else:
        DEV = "cuda:0"                               # noqa: N806
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        _install_gloo_cuda_shims()
    from bench import synth_records
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer
    from probabilisticteacher_amd.engine.flat import replicas_identical
    from probabilisticteacher_amd.modeling import sampling
    # global batch 2 + 2 (configs[3] in miniature: data/build.py:174-187 gives every rank total / world records per stream)
    cfg = setup_cfg("configs/pt/final_c2f.yaml", ["MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 1,
                                                  "SOLVER.IMG_PER_BATCH_LABEL", 2, "SOLVER.IMG_PER_BATCH_UNLABEL", 2])
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    gen = torch.Generator().manual_seed(770)
    glob = [tuple(synth_records(gen, 2, 320, 480, K, DEV) for _ in range(4)) for _ in range(3)]
    mine = [tuple(stream[rank:rank + 1] for stream in b) for b in glob]          # this rank's 1 + 1 share of every step

    class Recording(PTrainer):
        local = None

        def _write_metrics(self, record_dict, data_time, sumsq):          # this rank's own loss values, before the averaging
            self.local = {k: float(v.detach()) for k, v in record_dict.items() if k[:4] == "loss"}
            super()._write_metrics(record_dict, data_time, sumsq)

    def run(exchange):
        """3 steps (burn-in, EMA copy + mutual learning, mutual learning) with or without the cross-rank exchange"""
        torch.manual_seed(100 + rank)                # ranks start from DIFFERENT weights: the start-up broadcast must fix that
        ratios = iter([0.8, 0.6, 0.9, 0.7, 0.75, 0.65, 0.85, 0.55] * 3)
        tr = Recording(cfg, ratio_fn=lambda: next(ratios), grad_reduce=mode)
        assert tr.world_size == 2 and tr.reducer.active and len(tr.reducer.buckets) >= 4
        if not exchange:                             # a purely local trainer on the same start weights: hooks off, no averaging
            for h in tr.reducer._hooks:
                h.remove()
            tr.reducer.active = False
        keyg = torch.Generator().manual_seed(5 + rank)
        sampling.set_key_source(lambda labels, sizes, bg: torch.rand(labels.shape, generator=keyg))
        out = []
        try:
            for it in range(3):
                before = tr.student.flat.clone()
                m = tr.run_step(mine[it])
                out.append(dict(m=m, local=tr.local, grad=tr.student.grad.clone(), early=tr.reducer.launched_in_backward, before=before,
                                same=replicas_identical(tr.student.flat)[0] and replicas_identical(tr.teacher.flat)[0]))
                if not exchange:
                    break                            # without the exchange the replicas diverge after one step: compare step 0 only
        finally:
            sampling.set_key_source(None)
        return tr, out

    tr, ex = run(True)
    _, loc = run(False)
    # (ii) the exchanged gradient of step 0 == the mean of the two ranks' local gradients (same start weights, same data, same keys)
    local = loc[0]["grad"]
    both = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    mean = (both[0] + both[1]) / world
    err = float((ex[0]["grad"] - mean).abs().max() / mean.abs().max())
    # (iii) metrics: every rank reports rank 0's keys averaged over the ranks; each rank's own loss values (recorded before the averaging) give the expectation
    keys = sorted(ex[0]["local"])
    lm = torch.tensor([ex[0]["local"][k] for k in keys], dtype=torch.float64, device=DEV)
    lms = [torch.empty_like(lm) for _ in range(world)]
    dist.all_gather(lms, lm)
    q.put(dict(rank=rank, n_buckets=len(tr.reducer.buckets), early=[o["early"] for o in ex], same=[o["same"] for o in ex],
               start_equal=replicas_identical(ex[0]["before"])[0], grad_err=err,
               metrics=[{k: float(v) for k, v in o["m"].items()} for o in ex],
               local_loss_mean=((lms[0] + lms[1]) / world).cpu().numpy(), loss_keys=keys,
               local_differs=bool((lms[0] != lms[1]).any()),
               grad_head=np.asarray(ex[2]["grad"][-64:].cpu())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["all_reduce", "reduce_scatter"])
def test_real_run_step_on_two_ranks_sharing_the_gpu(mode):
    """Two PTrainer processes (gloo, world size 2) on cuda:0, each with half of a 2 + 2 batch, three real `run_step`s (burn-in,
    EMA copy + mutual learning, mutual learning) -- reference pt/engine/trainer.py:92-95 (DDP), :403-417 (metrics), :495-496
    (start-up sync), pt/data/build.py:174-187 (per-rank batch):
      (0) the start-up broadcast makes differently initialised ranks identical;
      (i)  student AND teacher replicas are bit-identical after every step (checksums, all-gathered);
      (ii) the gradient a rank holds after the exchange == the mean of the two ranks' local gradients (1e-6 of its scale);
      (iii) `_write_metrics`: both ranks report the same loss values = the mean of the ranks' local losses;
      (iv) all buckets but the last one leave DURING backward."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=900) for _ in procs], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, b = out
    for d in out:
        assert d["start_equal"], "start-up broadcast"
        assert all(d["same"]), f"replicas diverged: {d['same']}"
        assert d["grad_err"] <= 1e-6, f"exchanged gradient vs mean of local gradients: {d['grad_err']:.3e}"
        assert all(e >= d["n_buckets"] - 1 for e in d["early"]), (d["early"], d["n_buckets"])
    assert (a["grad_head"] == b["grad_head"]).all()
    for ma, mb in zip(a["metrics"], b["metrics"]):
        assert {k for k in ma if k[:4] == "loss"} == {k for k in mb if k[:4] == "loss"}
        for k in ma:
            if k[:4] == "loss" or k == "total_loss":
                assert ma[k] == mb[k] or (ma[k] != ma[k] and mb[k] != mb[k]), f"{k}: {ma[k]} vs {mb[k]}"
    assert a["local_differs"], "the two ranks see different data: their local losses must differ"
    for k, want in zip(a["loss_keys"], a["local_loss_mean"]):
        got = a["metrics"][0][k]
        assert abs(got - want) <= 1e-6 * abs(want) + 1e-7, f"step 0 {k}: reported {got} vs mean of the local losses {want}"


@pytest.mark.parametrize("mode", ["all_reduce", "reduce_scatter"])
def test_real_run_step_on_two_ranks_over_rccl(mode):
    """the same three-step, two-rank check as above on the `nccl` (= RCCL) backend with one GPU per rank -- the only
    configuration that takes the stream-ordered path of engine/flat.py (each bucket's in-place all-gather issued right behind its
    reduce-scatter, during backward).  Needs two visible GPUs: skipped, with the reason, on the 1-GPU boxes this repo is tested on."""
    if torch.cuda.device_count() < 2:
        pytest.skip(f"{torch.cuda.device_count()} GPU visible: the 2-rank RCCL run needs 2 (the gloo variant above runs everywhere)")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, mode, q, "nccl")) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=900) for _ in procs], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, b = out
    for d in out:
        assert d["start_equal"], "start-up broadcast"
        assert all(d["same"]), f"replicas diverged: {d['same']}"
        assert d["grad_err"] <= 1e-6, f"exchanged gradient vs mean of local gradients: {d['grad_err']:.3e}"
        assert all(e >= d["n_buckets"] - 1 for e in d["early"]), (d["early"], d["n_buckets"])
    assert (a["grad_head"] == b["grad_head"]).all()


def _two_rank_fullsize_worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_gloo_cuda_shims()
    from bench import synth_records
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer
    from probabilisticteacher_amd.engine.flat import replicas_identical
    B = 2                                            # per rank: 2 labelled + 2 unlabelled 1333 x 800 images (global 4 + 4)
    cfg = setup_cfg("configs/pt/final_c2f.yaml", ["MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 0,
                                                  "SOLVER.IMG_PER_BATCH_LABEL", B * world, "SOLVER.IMG_PER_BATCH_UNLABEL", B * world])
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    torch.manual_seed(0)
    tr = PTrainer(cfg, grad_reduce=mode)
    gen = torch.Generator().manual_seed(4242 + 1000 * rank)
    batch = tuple(synth_records(gen, B, 800, 1333, K, DEV) for _ in range(4))
    early, tails, tail_mb = [], [], []
    tr.reducer.diagnostics = True                    # (finish() measures tail_ms)
    for _ in range(3):
        tr.run_step(batch)
        early.append(tr.reducer.launched_in_backward)
        tails.append(tr.reducer.tail_ms)
        tail_mb.append(4e-6 * tr.reducer.tail_elems)
    sizes_mb = [4e-6 * (e - s) for s, e in tr.reducer.buckets]
    q.put(dict(rank=rank, n_buckets=len(tr.reducer.buckets), early=early, tails=tails, tail_mb=tail_mb, sizes_mb=sizes_mb,
               same=replicas_identical(tr.student.flat)[0] and replicas_identical(tr.teacher.flat)[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["all_reduce", "reduce_scatter"])
def test_two_ranks_at_1333x800_overlap_of_the_gradient_exchange(mode, capsys):
    """VERDICT r4 item 7 (multi-GPU readiness without hardware): the real mutual-learning `run_step` on two gloo ranks sharing cuda:0
    at the map size of configs[3] (1333 x 800; 2 + 2 images per rank instead of 8 + 8: the GPU is shared) -- reference
    pt/engine/trainer.py:92-95 (DDP), :384 (backward):
      * all buckets but the last one leave DURING backward, in each of three steps;
      * the last bucket (the first trainable layers) is at most 8 MB -- the only part of the exchange nothing can hide;
      * replicas bit-identical afterwards;
      * reported: bucket sizes and the host time between the end of backward and the completion of the last collective."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_fullsize_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=1500) for _ in procs], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for d in out:
        assert d["same"], "replicas diverged"
        assert all(e >= d["n_buckets"] - 1 for e in d["early"]), (d["early"], d["n_buckets"])
        assert d["sizes_mb"][-1] <= 8.0 + 1e-6 or mode == "reduce_scatter" and d["sizes_mb"][-1] <= 8.1, d["sizes_mb"]
        assert all(t <= 8.1 for t in d["tail_mb"]), d["tail_mb"]
    with capsys.disabled():
        d = out[0]
        print(f"\n[2 ranks, 1333x800, 2 + 2 per rank, {mode}] buckets MB {[round(v, 1) for v in d['sizes_mb']]}; launched during backward "
              f"{d['early']} of {d['n_buckets']}; un-overlapped tail per step: {[round(v, 2) for v in d['tail_mb']]} MB, "
              f"{[round(v, 1) for v in d['tails']]} ms host time (gloo through host copies on a shared GPU: an upper bound, not xGMI)")
