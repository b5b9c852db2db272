"""Shared helpers for the parity tests: rebuild the records a golden fixture describes."""
import os

import numpy as np
import torch

from oracle import d2, pt as opt

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def synth_image(seed, h, w):
    return torch.from_numpy(np.random.RandomState(int(seed)).randint(0, 256, (3, int(h), int(w))).astype(np.uint8))


def records(z, prefix, n, make_instances=None):
    """Rebuild `n` batch records stored by tools/gen_golden.py::records_to_arrays."""
    make_instances = make_instances or opt.FreeInstances
    out = []
    for i in range(n):
        if f"{prefix}{i}_imgseed" in z.files:
            s, gh, gw, fh, fw = [int(v) for v in z[f"{prefix}{i}_imgseed"]]
            img = synth_image(s, gh, gw)[:, :fh, :fw].contiguous()
        else:
            img = torch.from_numpy(z[f"{prefix}{i}_image"])
        h, w = img.shape[-2:]
        r = {"image": img, "height": h, "width": w}
        if f"{prefix}{i}_gt_boxes" in z.files:
            inst = make_instances((h, w))
            inst.gt_boxes = _boxes(make_instances, torch.from_numpy(z[f"{prefix}{i}_gt_boxes"]))
            inst.gt_classes = torch.from_numpy(z[f"{prefix}{i}_gt_classes"])
            r["instances"] = inst
        out.append(r)
    return out


def _boxes(make_instances, t):
    if make_instances is opt.FreeInstances:
        return d2.Boxes(t)
    from probabilisticteacher_amd.structures import Boxes
    return Boxes(t)


def close(a, b, rtol, atol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), f"{what}: max abs err {err.max():.3e} (worst rel {np.max(err / (np.abs(b) + 1e-30)):.3e})"
