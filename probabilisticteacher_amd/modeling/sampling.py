"""Label subsampling with an injectable permutation source (D2 subsample_labels).

The reference draws two `torch.randperm`s per image from the global RNG (rpn.py:433 via
RPN._subsample_labels; D2 StandardROIHeads._sample_proposals).  Parity tests inject the permutations through
`set_perm_fn`; production draws them on the device."""
from typing import Callable, Optional

import torch

_PERM_FN: Optional[Callable[[int], torch.Tensor]] = None


def set_perm_fn(fn: Optional[Callable[[int], torch.Tensor]]) -> None:
    global _PERM_FN
    _PERM_FN = fn


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
