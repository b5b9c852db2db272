"""Boxes / Instances / FreeInstances / ImageList containers (reference
pt/structures/instances.py:22-46).  Pure containers: all arithmetic on their tensors happens in HIP kernels."""
from __future__ import annotations

import itertools
from typing import Any, Dict, List, Tuple

import torch


class Boxes:
    """(N,4) xyxy fp32 boxes."""

    def __init__(self, tensor: torch.Tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.dtype != torch.float32:
            tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4))
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, *args, **kwargs):
        return Boxes(self.tensor.to(*args, **kwargs))

    def area(self):
        t = self.tensor
        return (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])

    def clip(self, box_size: Tuple[int, int]) -> None:
        h, w = box_size
        t = self.tensor
        self.tensor = torch.stack((t[:, 0].clamp(min=0, max=w), t[:, 1].clamp(min=0, max=h),
                                   t[:, 2].clamp(min=0, max=w), t[:, 3].clamp(min=0, max=h)), dim=-1)

    def nonempty(self, threshold: float = 0.0):
        t = self.tensor
        return ((t[:, 2] - t[:, 0]) > threshold) & ((t[:, 3] - t[:, 1]) > threshold)

    def scale(self, scale_x: float, scale_y: float) -> None:
        self.tensor = self.tensor * self.tensor.new_tensor([scale_x, scale_y, scale_x, scale_y])

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    @classmethod
    def cat(cls, boxes_list: List["Boxes"]):
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    @property
    def device(self):
        return self.tensor.device

    def __repr__(self):
        return f"Boxes({self.tensor})"


class Instances:
    """Attribute bag of per-instance fields with a shared length (detectron2.structures.Instances)."""

    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        self._image_size = image_size
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name, value):
        n = len(value)
        if len(self._fields):
            assert len(self) == n, f"Adding a field of length {n} to a Instances of length {len(self)}"
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def _new(self):
        return Instances(self._image_size)

    def to(self, *args, **kwargs):
        ret = self._new()
        for k, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(*args, **kwargs)
            ret.set(k, v)
        return ret

    def __getitem__(self, item):
        if type(item) == int:
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        raise NotImplementedError("Empty Instances does not support __len__!")

    @staticmethod
    def cat(instance_lists: List["Instances"]):
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        ret = Instances(instance_lists[0].image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = list(itertools.chain(*values))
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError(f"Unsupported type {type(v0)} for concatenation")
            ret.set(k, values)
        return ret


class FreeInstances(Instances):
    """Instances without the equal-length check (pt/structures/instances.py:22-46)."""

    def set(self, name, value):
        self._fields[name] = value

    def _new(self):
        return FreeInstances(self._image_size)


class ImageList:
    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    @property
    def device(self):
        return self.tensor.device
