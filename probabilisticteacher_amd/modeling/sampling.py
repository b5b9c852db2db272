"""Label subsampling with an injectable random-key source (D2 subsample_labels).

The reference draws two `torch.randperm`s per image from the global RNG (rpn.py:433 via
RPN._subsample_labels; D2 StandardROIHeads._sample_proposals): `pos[randperm(len(pos))[:k]]` needs len(pos) on the
host, i.e. two device->host syncs per image (128 per step at B = 16 + 16).

Here every candidate gets an i.i.d. random key and the k smallest keys win (== the first k entries of the permutation
argsort(keys[candidates])): the same uniformly random k-subsets, computed for the whole batch without a host sync.

The only injection point is the key source (`set_key_source`), the counterpart of seeding the reference's global RNG:
parity tests hand in keys that encode a given sequence of permutations (`perm_key_source` below for the permutations recorded
from the real reference, `keyed_perm_source` for a shared keyed stream), so the golden fixtures and the oracle comparisons
run through exactly this code."""
from typing import Callable, List, Optional

import torch

# fn(labels, sizes, bg_label) -> float32 keys in [0, 1) of labels' shape (any device).  `labels` is the (N, R) label
# matrix of a batch (sizes None) or the concatenation of per-image label vectors (sizes = their lengths).
KeySource = Callable[[torch.Tensor, Optional[List[int]], int], torch.Tensor]
_KEY_SOURCE: Optional[KeySource] = None
NOT_A_CANDIDATE = 2.0          # keys are in [0, 1)


def set_key_source(fn: Optional[KeySource]) -> None:
    """None = torch.rand on the device (production)."""
    global _KEY_SOURCE
    _KEY_SOURCE = fn


def perm_key_source(perm_fn) -> KeySource:
    """Keys that make the keyed sampler select exactly `pos[perm_fn(len(pos))[:k]]`, `neg[perm_fn(len(neg))[:k]]` -- image by
    image, positives first: the order in which the reference draws its two `randperm`s (D2 subsample_labels).  Candidate j of
    a permutation gets key (rank + .5) / (n + 1), so ascending key order == permutation order.  `perm_fn(n)` returns a
    permutation of range(n): a replay of recorded `randperm` results, or a seeded generator -- the counterpart of seeding the
    reference's global RNG."""
    def src(labels, sizes, bg_label):
        lab = labels.detach().cpu()
        keys = torch.zeros(lab.shape, dtype=torch.float32)
        if sizes is None:
            rows, krows = list(lab), list(keys)
        else:
            rows, krows = list(torch.split(lab, [int(s) for s in sizes])), list(torch.split(keys, [int(s) for s in sizes]))
        for lb, k in zip(rows, krows):
            for idx in (torch.nonzero((lb != -1) & (lb != bg_label)).squeeze(1), torch.nonzero(lb == bg_label).squeeze(1)):
                p = perm_fn(int(idx.numel()))
                k[idx[p]] = (torch.arange(idx.numel(), dtype=torch.float32) + 0.5) / (idx.numel() + 1)
        return keys
    return src


def keyed_perm_source(kp) -> KeySource:
    """A keyed-permutation object (`kp.draw(shape)` -> uniform keys, e.g. a seeded stream shared with another implementation)
    as the key source: one key row per image, in image order."""
    def src(labels, sizes, bg_label):
        if sizes is None:
            return kp.draw(tuple(labels.shape))
        return torch.cat([kp.draw((int(s),)) for s in sizes]) if len(sizes) else torch.zeros(0)
    return src


def draw_keys(labels: torch.Tensor, sizes: Optional[List[int]], bg_label: int) -> torch.Tensor:
    if _KEY_SOURCE is not None:
        keys = _KEY_SOURCE(labels, sizes, bg_label).to(device=labels.device, dtype=torch.float32)
        assert keys.shape == labels.shape
        return keys
    return torch.rand(labels.shape, device=labels.device)


def keyed_topk(mask: torch.Tensor, keys: torch.Tensor, k: int):
    """The (at most k) candidates of `mask` with the smallest keys, along the last dim, in ascending key order:
    (indices (..., k'), valid (..., k') bool) with the valid entries first.  No host sync."""
    k = min(k, mask.shape[-1])
    vals, idx = torch.topk(torch.where(mask, keys, keys.new_full((), NOT_A_CANDIDATE)), k, dim=-1, largest=False,
                           sorted=True)
    return idx, vals < NOT_A_CANDIDATE


def keyed_relabel(labels: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int) -> torch.Tensor:
    """RPN._subsample_labels (D2) for a whole batch: labels (N, R) int8 in {-1, bg, fg...} ->
    new labels with the sampled positives = 1, sampled negatives = 0, everything else -1.  No host sync."""
    n, r = labels.shape
    keys = draw_keys(labels, None, bg_label)
    if labels.is_cuda:                 # the product path: one radix-select kernel (the torch form below is the host-logic
        from .. import ops             # statement the CPU tests pin and the GPU test compares the kernel with)
        return ops.rpn_subsample_relabel(labels, keys, num_samples, int(num_samples * positive_fraction), bg_label)
    pos_m = (labels != -1) & (labels != bg_label)
    neg_m = labels == bg_label
    ip, vp = keyed_topk(pos_m, keys, int(num_samples * positive_fraction))
    ineg, vn = keyed_topk(neg_m, keys, num_samples)
    num_pos = vp.sum(dim=1, keepdim=True)
    vn = vn & (torch.arange(ineg.shape[1], device=labels.device).unsqueeze(0) < (num_samples - num_pos))
    out = torch.full_like(labels, -1)
    # entries that are not selected rewrite the value already there (top-k indices are unique per row)
    out.scatter_(1, ip, torch.where(vp, torch.ones_like(ip, dtype=labels.dtype), out.gather(1, ip)))
    out.scatter_(1, ineg, torch.where(vn, torch.zeros_like(ineg, dtype=labels.dtype), out.gather(1, ineg)))
    return out
