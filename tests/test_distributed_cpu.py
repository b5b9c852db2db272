"""CPU (gloo, world_size 2): the data-parallel exchange of the step -- flat gradient all-reduce (DDP average),
parameter broadcast, and the 1-rank-batch-2B == 2-ranks-batch-B gradient equivalence."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from probabilisticteacher_amd.engine.flat import FlatParams, allreduce_mean_, broadcast_
    torch.manual_seed(100 + rank)                         # different init per rank ...
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    model[0].bias.requires_grad_(False)                   # a frozen tensor must sort after the trainable ones
    flat = FlatParams(model)
    broadcast_(flat.flat)                                 # ... made identical by the startup broadcast
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 6, generator=g)
    y = torch.randn(8, 3, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]   # rank-local half batch
    flat.zero_grad()
    ((model(xs) - ys) ** 2).mean().backward()
    allreduce_mean_(flat.grad, world, chunk_elems=7)      # tiny chunks to exercise the chunking
    # numpy payloads: a torch tensor on an mp.Queue travels as a shared-memory fd that the parent may open only after this
    # worker has exited (FileNotFoundError in rebuild_storage_fd, seen 1 in 2 runs)
    q.put((rank, flat.flat.detach().clone().numpy(), flat.grad.detach().clone().numpy(), flat.n_trainable, list(flat.index.keys())))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_matches_single_rank_big_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, g0, nt0, names0), (_, p1, g1, nt1, names1) = [(r, torch.from_numpy(p), torch.from_numpy(g), nt, nm) for r, p, g, nt, nm in res]
    assert torch.equal(p0, p1), "broadcast must make the parameters identical"
    assert torch.equal(g0, g1), "all ranks hold the same averaged gradient"
    assert names0 == names1 and names0[-1] == "0.bias", "frozen parameters are laid out after the trainable ones"
    # single process, full batch, same parameters: the mean-of-means over equal shards == the full-batch mean
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    model[0].bias.requires_grad_(False)
    from probabilisticteacher_amd.engine.flat import FlatParams
    flat = FlatParams(model)
    flat.flat.copy_(p0)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 6, generator=g)
    y = torch.randn(8, 3, generator=g)
    flat.zero_grad()
    ((model(x) - y) ** 2).mean().backward()
    assert flat.n_trainable == nt0
    assert torch.allclose(flat.grad, g0, rtol=1e-5, atol=1e-7)


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.unused = torch.nn.Parameter(torch.ones(5))           # trainable, never gets a gradient; FIRST in the layout
        self.body = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                                        torch.nn.Linear(16, 3))

    def forward(self, x):
        return self.body(x)


def _mlp():
    return _Net()


def _worker_bucketed(rank, world, port, q, mode="all_reduce"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from probabilisticteacher_amd.engine.flat import BucketedGradReducer, FlatParams, allreduce_mean_, broadcast_
    torch.manual_seed(3 + rank)
    model, ref = _mlp(), _mlp()
    flat, rflat = FlatParams(model), FlatParams(ref)
    broadcast_(flat.flat)
    rflat.flat.copy_(flat.flat)
    red = BucketedGradReducer(flat, world, bucket_elems=100, mode=mode)      # several buckets, tail of the buffer first
    g = torch.Generator().manual_seed(20 + rank)                  # rank-local data
    res = []
    for step in range(2):                                         # two steps: the reducer re-arms itself
        x, y = torch.randn(4, 6, generator=g), torch.randn(4, 3, generator=g)
        flat.zero_grad()
        ((model(x) - y) ** 2).mean().backward()                   # hooks launch ready buckets during backward
        launched = red.next
        got = red.finish().clone()
        rflat.zero_grad()
        ((ref(x) - y) ** 2).mean().backward()
        want = allreduce_mean_(rflat.grad, world).clone()
        res.append((got.numpy(), want.numpy(), launched))
    # a second backward() without finish() / zero_grad() in between would write into buckets whose collectives are in
    # flight -- the hook refuses (same on every rank, so nobody is left waiting); zero_grad() then drains and re-arms
    flat.zero_grad()
    ((model(x) - y) ** 2).mean().backward()
    try:
        ((model(x) - y) ** 2).mean().backward()
        refused = False
    except RuntimeError as e:
        refused = "after its collective was launched" in str(e)
    flat.zero_grad()
    ((model(x) - y) ** 2).mean().backward()
    red.finish()
    q.put((rank, res, list(red.buckets), red.bucket_of["unused"], refused))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("mode", ["all_reduce", "reduce_scatter"])
def test_bucketed_reducer_overlaps_and_matches_plain_allreduce(mode):
    """mode: one all-reduce per bucket, or reduce-scatter + all-gather on shard-aligned buckets (engine/flat.py)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bucketed, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res, buckets, unused_bucket, refused in out:
        assert refused, "a second backward before finish() must be refused by the bucket hooks"
        assert len(buckets) >= 3 and buckets[0][1] == max(b[1] for b in buckets), "bucket 0 is the tail of the buffer"
        assert sorted(buckets)[0][0] == 0 and all(a[0] == b[1] for a, b in zip(buckets[:-1], buckets[1:])), "contiguous cover"
        for got, want, launched in res:
            assert torch.allclose(torch.from_numpy(got), torch.from_numpy(want), rtol=1e-6, atol=1e-8), "bucketed exchange == plain mean all-reduce"
            # every bucket before the one holding the gradient-less parameter went out DURING backward
            assert launched == unused_bucket == len(buckets) - 1 and launched >= 2, (launched, unused_bucket)
    assert (out[0][1][1][0] == out[1][1][1][0]).all(), "both ranks hold the same averaged gradient"


def _worker_metrics(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    from probabilisticteacher_amd.engine.trainer import PTrainer
    me = SimpleNamespace(world_size=world, METRIC_KEYS=PTrainer.METRIC_KEYS, last_metrics={})
    if rank == 0:
        rec = {"loss_cls_sup": torch.tensor(1.0), "loss_rpn_cls_sup": torch.tensor(3.0), "loss_cls_unsup": torch.tensor(5.0)}
    else:                                                     # a different key set on the other rank
        rec = {"loss_cls_sup": torch.tensor(2.0), "loss_box_reg_sup": torch.tensor(7.0)}
    PTrainer._write_metrics(me, rec, 0.25 + rank, torch.tensor([4.0 * (rank + 1) ** 2]))
    q.put((rank, me.last_metrics))
    dist.barrier()
    dist.destroy_process_group()


def test_write_metrics_reference_semantics_with_rank_dependent_keys():
    """reference trainer.py:403-417: rank 0's keys, each the mean over ALL ranks with 0.0 where a rank lacks the key;
    data_time = max over ranks"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_metrics, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m0 = out[0]
    assert set(m0) == {"loss_cls_sup", "loss_rpn_cls_sup", "loss_cls_unsup", "total_loss", "grad_norm", "data_time"}
    assert m0["loss_cls_sup"] == 1.5 and m0["loss_rpn_cls_sup"] == 1.5 and m0["loss_cls_unsup"] == 2.5
    assert m0["total_loss"] == 5.5 and m0["data_time"] == 1.25 and m0["grad_norm"] == 2.0 and out[1]["grad_norm"] == 4.0


def _worker_checksum(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from probabilisticteacher_amd.engine.flat import BucketedGradReducer, FlatParams, broadcast_, replicas_identical
    torch.manual_seed(3 + rank)
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    flat = FlatParams(model)
    same_before, _ = replicas_identical(flat.flat)            # different seeds: replicas differ
    broadcast_(flat.flat)
    red = BucketedGradReducer(flat, world, bucket_elems=20)
    g = torch.Generator().manual_seed(50 + rank)                # rank-local data
    for _ in range(3):                                          # three data-parallel SGD steps
        flat.zero_grad()
        ((model(torch.randn(4, 6, generator=g)) - torch.randn(4, 3, generator=g)) ** 2).mean().backward()
        grad = red.finish()
        with torch.no_grad():
            flat.trainable().add_(grad, alpha=-0.1)
    same_after, sums = replicas_identical(flat.flat)
    if rank == 1:
        with torch.no_grad():
            flat.flat[5] += 1e-6                                # one rank drifts by one ulp-ish step
    same_drift, _ = replicas_identical(flat.flat)
    q.put((rank, same_before, same_after, same_drift, sums, red.bytes_per_step, flat.n_trainable))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_checksums_detect_divergence_across_ranks():
    """what bench.py asserts after the timed steps at N > 1: data-parallel replicas stay bit-identical (int64 sum of the
    parameters' bit patterns, one all-gather) -- and the check really fires when one rank drifts"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_checksum, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, before, after, drift, sums, nbytes, nt in out:
        assert not before and after and not drift, (rank, before, after, drift)
        assert len(sums) == 2 and sums[0] == sums[1]
        assert nbytes == 4 * nt, "every trainable element is exchanged exactly once per step"

