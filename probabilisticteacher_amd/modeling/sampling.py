"""Label subsampling with an injectable permutation source (D2 subsample_labels).

The reference draws two `torch.randperm`s per image from the global RNG (rpn.py:433 via
RPN._subsample_labels; D2 StandardROIHeads._sample_proposals): `pos[randperm(len(pos))[:k]]` needs len(pos) on the
host, i.e. two device->host syncs per image (128 per step at B = 16 + 16).

Production path (`keyed_*`): the same uniformly random k-subsets without any host sync -- every candidate gets an
i.i.d. random key and the k smallest keys win (== the first k entries of the permutation argsort(keys[candidates])).
Parity tests either inject the reference's permutations through `set_perm_fn` (legacy per-image path, used with the
golden fixtures) or inject the keys through `set_key_fn` (the oracle then derives its permutations from the same
keys, oracle/pt.py KeyedPerm)."""
from typing import Callable, Optional

import torch

_PERM_FN: Optional[Callable[[int], torch.Tensor]] = None
_KEY_FN: Optional[Callable[[tuple], torch.Tensor]] = None
NOT_A_CANDIDATE = 2.0          # keys are in [0, 1)


def set_perm_fn(fn: Optional[Callable[[int], torch.Tensor]]) -> None:
    global _PERM_FN
    _PERM_FN = fn


def set_key_fn(fn: Optional[Callable[[tuple], torch.Tensor]]) -> None:
    """fn(shape) -> float32 CPU tensor of keys in [0, 1); None = torch.rand on the device."""
    global _KEY_FN
    _KEY_FN = fn


def legacy_path() -> bool:
    return _PERM_FN is not None


def _keys(shape, device) -> torch.Tensor:
    if _KEY_FN is not None:
        return _KEY_FN(tuple(shape)).to(device)
    return torch.rand(shape, device=device)


def segment_keys(sizes, device) -> torch.Tensor:
    """One key per element of the concatenated label vectors; with an injected key source the keys are drawn image
    by image (the order the oracle's KeyedPerm replays them)."""
    if _KEY_FN is not None:
        return torch.cat([_KEY_FN((int(n),)) for n in sizes]).to(device) if len(sizes) else torch.zeros(0, device=device)
    return torch.rand(int(sum(sizes)), device=device)


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
    keys = _keys((n, r), labels.device)
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


def keyed_sample(cls: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int):
    """D2 subsample_labels for one image without a host sync: (fg_idx, n_fg, bg_idx, n_bg) where the first n_fg
    (n_bg) entries of fg_idx (bg_idx) are the sample, in the order the reference's permutation would give; n_* are
    0-dim device tensors."""
    keys = _keys((cls.shape[0],), cls.device)
    i_f, v_f = keyed_topk((cls != -1) & (cls != bg_label), keys, int(num_samples * positive_fraction))
    i_b, v_b = keyed_topk(cls == bg_label, keys, num_samples)
    n_f = v_f.sum()
    n_b = torch.minimum(v_b.sum(), num_samples - n_f)
    return i_f, n_f, i_b, n_b


def _perm(n: int, device) -> torch.Tensor:
    if _PERM_FN is not None:
        return _PERM_FN(n).to(device)
    return torch.randperm(n, device=device)


def subsample_labels(labels: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int):
    positive = torch.nonzero((labels != -1) & (labels != bg_label)).squeeze(1)
    negative = torch.nonzero(labels == bg_label).squeeze(1)
    num_pos = min(positive.numel(), int(num_samples * positive_fraction))
    num_neg = min(negative.numel(), num_samples - num_pos)
    perm1 = _perm(positive.numel(), labels.device)[:num_pos]
    perm2 = _perm(negative.numel(), labels.device)[:num_neg]
    return positive[perm1], negative[perm2]
