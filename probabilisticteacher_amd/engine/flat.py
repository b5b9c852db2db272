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
        order = [(n, p) for n, p in named if p.requires_grad] + [(n, p) for n, p in named if not p.requires_grad]
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


def broadcast_(flat: torch.Tensor, src: int = 0, group=None):
    """trainer.py:495 `_sync_params_and_buffers`: rank 0's parameters to everyone, one collective."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


def lr_at(cfg, it: int) -> float:
    """D2 WarmupMultiStepLR: BASE_LR * warmup(it) * GAMMA^bisect_right(STEPS, it)."""
    import bisect

    S = cfg.SOLVER
    if it >= S.WARMUP_ITERS:
        f = 1.0
    elif S.WARMUP_METHOD == "constant":
        f = S.WARMUP_FACTOR
    else:
        alpha = it / S.WARMUP_ITERS
        f = S.WARMUP_FACTOR * (1 - alpha) + alpha
    return S.BASE_LR * f * S.GAMMA ** bisect.bisect_right(list(S.STEPS), it)
