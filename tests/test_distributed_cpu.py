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
    q.put((rank, flat.flat.clone(), flat.grad.clone(), flat.n_trainable, list(flat.index.keys())))
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
    (_, p0, g0, nt0, names0), (_, p1, g1, nt1, names1) = res
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
