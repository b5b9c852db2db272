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

    def attach_grads(self) -> torch.Tensor:
        """Point every trainable parameter's .grad at a view of one flat buffer (autograd accumulates in place)."""
        if self.grad is None:
            self.grad = torch.zeros(self.n_trainable, dtype=torch.float32, device=self.flat.device)
        for n, p in self.params.items():
            if p.requires_grad:
                off, k = self.index[n]
                p.grad = self.grad[off:off + k].view(p.shape)
        return self.grad

    def zero_grad(self) -> None:
        self.attach_grads().zero_()
        for fn in self.on_zero_grad:
            fn()

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
                 force: bool = False):
        """force: install the hooks and run the collectives even for world_size 1 (the sum over one rank is the identity;
        used to exercise the RCCL / stream-ordering path on a single GPU)."""
        self.fp, self.world, self.group = fp, world_size, group
        self.active = world_size > 1 or force
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
        self.size = [sum(1 for b in self.bucket_of.values() if b == i) for i in range(len(self.buckets))]
        self._grad = grad
        self._hooks = []
        if self.active:
            for n in names:
                p = fp.params[n]
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(self.bucket_of[n])))
        self.reset()
        # re-arm at the start of every step: a backward() that was not followed by finish() (an exception mid-step, a
        # test) must not leave stale ready-counts behind
        fp.on_zero_grad.append(self.reset)

    def reset(self):
        for h in getattr(self, "handles", []):      # collectives of an abandoned step: drain before re-arming
            h.wait()
        self.pending = list(self.size)
        self.next = 0
        self.handles = []
        self.launched_in_backward = 0               # (diagnostic) buckets that left before finish()

    def _make_hook(self, b: int):
        def hook(_param):
            self.pending[b] -= 1
            self._launch_ready()
        return hook

    def _launch(self, b: int):
        s, e = self.buckets[b]
        self.handles.append(dist.all_reduce(self._grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _launch_ready(self):
        while self.next < len(self.buckets) and self.pending[self.next] <= 0:
            self._launch(self.next)
            self.next += 1
            self.launched_in_backward += 1

    def finish(self) -> torch.Tensor:
        """Call after backward(): launches the remaining buckets in order, waits for all, averages."""
        if self.active:
            while self.next < len(self.buckets):
                self._launch(self.next)
                self.next += 1
            for h in self.handles:
                h.wait()
            self.handles = []
            if self.world > 1:
                self._grad.div_(self.world)
        n_early = self.launched_in_backward
        self.reset()
        self.launched_in_backward = n_early
        return self._grad


def broadcast_(flat: torch.Tensor, src: int = 0, group=None):
    """trainer.py:495 `_sync_params_and_buffers`: rank 0's parameters to everyone, one collective."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


from ..solver import lr_at  # noqa: E402,F401  (re-exported: the schedule lives in solver.py)
