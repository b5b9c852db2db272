"""Flat parameter / gradient / momentum buffers and the data-parallel gradient exchange.

MI355X-first layout: every parameter of a model is a view into ONE contiguous fp32 buffer (trainable parameters
first, frozen ones after), gradients and SGD momentum live in equally laid-out buffers.  That turns
  * the EMA teacher update (reference trainer.py:431-449: ~40 per-tensor kernel chains + load_state_dict),
  * gradient clipping (trainer.py:592-603: per-parameter norms + .item() + per-parameter mul_) and
  * the SGD step (torch.optim.SGD per-parameter loops)
into one HBM-bound launch each, and the DDP gradient all-reduce (trainer.py:92-95) into a few large RCCL
collectives on the flat gradient buffer (sized for xGMI: big messages, no per-tensor latency)."""
from collections import OrderedDict
from typing import Dict, List, Tuple

import time

import torch
import torch.distributed as dist
from torch import nn


class FlatParams:
    def __init__(self, model: nn.Module):
        named = list(model.named_parameters())
        # trainable parameters first; among them the learnable anchor table leads: it is the one parameter that gets no
        # gradient in supervised-only steps (anchors are detached unless danchor=True), and the bucketed reducer below
        # walks the buffer from its END, so a never-ready parameter must not sit in front of the backbone's buckets
        train = [(n, p) for n, p in named if p.requires_grad]
        train = [(n, p) for n, p in train if "anchor_generator" in n] + [(n, p) for n, p in train if "anchor_generator" not in n]
        order = train + [(n, p) for n, p in named if not p.requires_grad]
        total = sum(p.numel() for _, p in order)
        dev = order[0][1].device
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.index: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
        off = 0
        self.n_trainable = 0
        for n, p in order:
            k = p.numel()
            view = self.flat[off:off + k].view(p.shape)
            view.copy_(p.data)
            p.data = view
            self.index[n] = (off, k)
            off += k
            if p.requires_grad:
                self.n_trainable = off
        self.params = OrderedDict(order)
        self.grad = None
        self.on_zero_grad = []          # callbacks run at the start of every step (the gradient reducer re-arms itself)

    GRAD_PAD = 64       # the gradient storage is padded to a multiple of this (reduce-scatter shards: any world size dividing 64)

    def attach_grads(self) -> torch.Tensor:
        """Point every trainable parameter's .grad at a view of one flat buffer (autograd accumulates in place)."""
        if self.grad is None:
            padded = (self.n_trainable + self.GRAD_PAD - 1) // self.GRAD_PAD * self.GRAD_PAD
            self.grad_storage = torch.zeros(padded, dtype=torch.float32, device=self.flat.device)
            self.grad = self.grad_storage[: self.n_trainable]
        for n, p in self.params.items():
            if p.requires_grad:
                off, k = self.index[n]
                p.grad = self.grad[off:off + k].view(p.shape)
        return self.grad

    def zero_grad(self) -> None:
        # callbacks first: the gradient reducer waits for collectives an abandoned step may have left in flight -- they
        # write into this buffer, so it is zeroed only after they have drained
        for fn in self.on_zero_grad:
            fn()
        self.attach_grads().zero_()

    def trainable(self) -> torch.Tensor:
        return self.flat[: self.n_trainable]


def allreduce_mean_(flat_grad: torch.Tensor, world_size: int, chunk_elems: int = 16 * 1024 * 1024, group=None):
    """Average `flat_grad` over ranks in place: DDP's gradient all-reduce (trainer.py:92-95,384) issued as a few
    large collectives over the flat buffer.  `backend="nccl"` is RCCL over xGMI on ROCm; gloo works for CPU tests."""
    if world_size <= 1:
        return flat_grad
    handles = []
    for s in range(0, flat_grad.numel(), chunk_elems):
        handles.append(dist.all_reduce(flat_grad[s:s + chunk_elems], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for h in handles:
        h.wait()
    flat_grad.div_(world_size)
    return flat_grad


class BucketedGradReducer:
    """DDP's overlapped gradient exchange on the flat buffer: the flat gradient is cut into contiguous buckets, walking
    from its END (the box head, whose gradients are produced first in backward, lies last), and each bucket's
    all-reduce (RCCL over xGMI, async on the collective stream) is launched from a post-accumulate-grad hook as soon as
    every parameter in it has its gradient -- while the backbone's dgrad / wgrad kernels are still running.

    Buckets are always launched in bucket order (a ready bucket waits for its predecessors), so every rank issues the
    same sequence of collectives even if autograd visits parameters in a different order; `finish()` launches what is
    left (parameters that received no gradient this step keep their zeros), waits, and divides by the world size."""

    def __init__(self, fp: FlatParams, world_size: int, bucket_elems: int = 16 * 1024 * 1024, group=None,
                 force: bool = False, mode: str = "all_reduce", last_bucket_elems: int = 2 * 1024 * 1024):
        """force: install the hooks and run the collectives even for world_size 1 (the sum over one rank is the identity;
        used to exercise the RCCL / stream-ordering path on a single GPU).
        mode: "all_reduce" -- one all-reduce per bucket; "reduce_scatter" -- the same exchange spelled as a reduce-scatter
        per bucket during backward + the buckets' all-gathers in finish() (on the fully connected xGMI mesh each rank then owns 1/W of a bucket and the
        two halves use all seven links; which of the two RCCL runs faster is a measurement for the first 8-GPU node)."""
        if mode not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"unknown gradient exchange {mode!r}")
        self.fp, self.world, self.group, self.mode = fp, world_size, group, mode
        self.active = world_size > 1 or force
        self.diagnostics = False        # True: finish() also measures tail_ms (host clock; tests / tools set it, the step does not pay for it)
        # collectives of one group complete in issue order (RCCL's stream): the reduce-scatter mode can then issue each bucket's
        # all-gather right behind its reduce-scatter
        self.stream_ordered = bool(dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl")
        self.gather_handles: List = []
        grad = fp.attach_grads()
        names = [n for n, p in fp.params.items() if p.requires_grad]
        self.buckets: List[Tuple[int, int]] = []         # (start, end) in elements, bucket 0 = tail of the buffer
        self.bucket_of: Dict[str, int] = {}
        end = fp.n_trainable
        members: List[str] = []
        for n in reversed(names):
            off, k = fp.index[n]
            members.append(n)
            if end - off >= bucket_elems or off == 0:
                for m in members:
                    self.bucket_of[m] = len(self.buckets)
                self.buckets.append((off, end))
                end, members = off, []
        # the LAST bucket (the first trainable layers: their gradients are ready when backward ends, nothing is left to hide its
        # exchange behind) is at most last_bucket_elems (8 MB): cut at the highest parameter boundary that keeps it so
        if self.buckets and self.buckets[-1][1] - self.buckets[-1][0] > last_bucket_elems:
            s0, e0 = self.buckets[-1]
            b_last = len(self.buckets) - 1
            bounds = sorted(fp.index[n][0] for n in names if s0 < fp.index[n][0] < e0)
            cut = max([b for b in bounds if b - s0 <= last_bucket_elems], default=None)
            if cut is not None:
                self.buckets[-1] = (cut, e0)
                self.buckets.append((s0, cut))
                for n in names:
                    if self.bucket_of.get(n) == b_last and fp.index[n][0] < cut:
                        self.bucket_of[n] = b_last + 1
        if mode == "reduce_scatter":
            # bucket boundaries on multiples of the world size (equal shards): a cut moves UP to the next multiple, i.e. the
            # first elements of a bucket's lowest parameter travel with the following (later) bucket -- which therefore also
            # waits for that parameter; the buffer's padded tail belongs to bucket 0
            w = max(world_size, 1)
            assert fp.GRAD_PAD % w == 0, f"world size {w} must divide {fp.GRAD_PAD}"
            cuts = [min((s + w - 1) // w * w, fp.grad_storage.numel()) for s, _ in self.buckets]
            ends = [fp.grad_storage.numel()] + cuts[:-1]
            self.buckets = [(s, e) for s, e in zip(cuts, ends)]
            assert self.buckets[-1][0] == 0 and all(e > s for s, e in self.buckets)
            names_in = {i: set() for i in range(len(self.buckets))}
            for n in names:
                off, k = fp.index[n]
                for i, (s, e) in enumerate(self.buckets):
                    if off < e and off + k > s:
                        names_in[i].add(n)
            self.members = names_in
            self.shards = [torch.empty((e - s) // w, dtype=torch.float32, device=grad.device) for s, e in self.buckets]
        else:
            self.members = {i: {n for n, b in self.bucket_of.items() if b == i} for i in range(len(self.buckets))}
        self.size = [len(self.members[i]) for i in range(len(self.buckets))]
        self.bytes_per_step = 4 * sum(e - s for s, e in self.buckets)
        self._grad = grad
        self._hooks = []
        if self.active:
            for n in names:
                p = fp.params[n]
                self._hooks.append(p.register_post_accumulate_grad_hook(
                    self._make_hook([i for i in range(len(self.buckets)) if n in self.members[i]])))
        self.reset()
        # re-arm at the start of every step: a backward() that was not followed by finish() (an exception mid-step, a
        # test) must not leave stale ready-counts behind
        fp.on_zero_grad.append(self.reset)

    def reset(self):
        for h in getattr(self, "handles", []):      # collectives of an abandoned step: drain before re-arming
            h.wait()
        for h in getattr(self, "gather_handles", []):
            h.wait()
        self.pending = list(self.size)
        self.next = 0
        self.handles = []
        self.gather_handles = []
        self.launched_in_backward = 0               # (diagnostic) buckets that left before finish()
        self.tail_ms, self.tail_elems = getattr(self, "tail_ms", 0.0), getattr(self, "tail_elems", 0)

    def _make_hook(self, bs):
        def hook(_param):
            for b in bs:
                # a bucket that has left must never be touched again: its collective (and, on RCCL, the in-place all-gather) owns
                # grad_storage[s:e] until finish().  One backward() per step keeps this true; a second backward or gradient
                # accumulation without finish() / zero_grad() in between would not -- fail loudly instead of racing
                if b < self.next:
                    raise RuntimeError(f"gradient bucket {b} received another gradient after its collective was launched: "
                                       "call finish() (or zero_grad()) between two backward passes")
                self.pending[b] -= 1
            self._launch_ready()
        return hook

    def _launch(self, b: int):
        s, e = self.buckets[b]
        if self.mode == "all_reduce":
            self.handles.append(dist.all_reduce(self._grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:       # first half now (hidden behind the rest of backward)
            buf = self.fp.grad_storage[s:e]
            self.handles.append(dist.reduce_scatter_tensor(self.shards[b], buf, op=dist.ReduceOp.SUM, group=self.group,
                                                           async_op=True))
            if self.stream_ordered:
                # RCCL runs one group's collectives in issue order on its stream: the all-gather can follow at once and overlaps
                # with backward as well.  gloo's worker threads do not keep that order (a world-size-2 gloo test caught
                # the all-gather overtaking): there it is issued in finish(), after the reduce-scatters have completed
                self.gather_handles.append(dist.all_gather_into_tensor(buf, self.shards[b], group=self.group, async_op=True))

    def _launch_ready(self):
        while self.next < len(self.buckets) and self.pending[self.next] <= 0:
            self._launch(self.next)
            self.next += 1
            self.launched_in_backward += 1

    def finish(self) -> torch.Tensor:
        """Call after backward(): launches the remaining buckets in order, waits for all, averages."""
        if self.active:
            t0 = time.perf_counter() if self.diagnostics else 0.0
            self.tail_elems = sum(e - s for s, e in self.buckets[self.next:])     # (diagnostic) what could not leave during backward
            while self.next < len(self.buckets):
                self._launch(self.next)
                self.next += 1
            for h in self.handles:
                h.wait()
            self.handles = []
            if self.mode == "reduce_scatter":       # second half: every rank's reduced shard back into the flat buffer
                hs = self.gather_handles if self.stream_ordered else [
                    dist.all_gather_into_tensor(self.fp.grad_storage[s:e], self.shards[b], group=self.group, async_op=True)
                    for b, (s, e) in enumerate(self.buckets)]
                for h in hs:
                    h.wait()
                self.gather_handles = []
            if self.world > 1:
                self._grad.div_(self.world)
            # (diagnostic) host time from the end of backward to the last collective's completion handle: with gloo (blocking
            # waits) the exposed tail of the exchange; with RCCL the waits only order streams
            if self.diagnostics:
                self.tail_ms = 1e3 * (time.perf_counter() - t0)
        n_early = self.launched_in_backward
        self.reset()
        self.launched_in_backward = n_early
        return self._grad


def replica_checksum(flat: torch.Tensor) -> torch.Tensor:
    """Order-independent exact checksum of a flat fp32 buffer: the int64 sum of its bit patterns (one device scalar)."""
    return flat.view(torch.int32).to(torch.int64).sum()


def replicas_identical(flat: torch.Tensor, group=None):
    """Data parallelism keeps the replicas bit-identical: every rank applies the same averaged gradient to the same
    parameters (reference trainer.py:92-95; no parameter broadcast after start-up).  Returns (all equal?, per-rank checksums);
    one all-gather of one int64 per rank."""
    mine = replica_checksum(flat).reshape(1)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return True, [int(mine.item())]
    allv = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(allv, mine, group=group)
    vals = [int(v.item()) for v in allv]
    return all(v == vals[0] for v in vals), vals


def broadcast_(flat: torch.Tensor, src: int = 0, group=None):
    """trainer.py:495 `_sync_params_and_buffers`: rank 0's parameters to everyone, one collective."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


from ..solver import lr_at  # noqa: E402,F401  (re-exported: the schedule lives in solver.py)
