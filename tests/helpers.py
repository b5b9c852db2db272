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
    if os.environ.get("PTMI_TEST_VERBOSE") and err.size:
        print(f"[close] {what}: max abs err {err.max():.3e}, max tol-ratio {np.max(err / tol):.3f}")
    assert (err <= tol).all(), f"{what}: max abs err {err.max():.3e} (worst rel {np.max(err / (np.abs(b) + 1e-30)):.3e})"


def match_detections(boxes_a, cls_a, boxes_b, cls_b, box_tol=5e-3):
    """Order-insensitive comparison of two detection lists (scores that differ by fp32 noise may swap ranks):
    returns (fraction of b matched by a same-class box within box_tol px, index map b -> a or -1)."""
    boxes_a, boxes_b = np.asarray(boxes_a, np.float64), np.asarray(boxes_b, np.float64)
    cls_a, cls_b = np.asarray(cls_a), np.asarray(cls_b)
    used = np.zeros(len(boxes_a), bool)
    idx = -np.ones(len(boxes_b), np.int64)
    for j in range(len(boxes_b)):
        cand = np.nonzero((cls_a == cls_b[j]) & ~used)[0]
        if len(cand) == 0:
            continue
        d = np.abs(boxes_a[cand] - boxes_b[j]).max(axis=1)
        k = int(np.argmin(d))
        if d[k] <= box_tol * max(1.0, np.abs(boxes_b[j]).max() / 100.0):
            idx[j] = cand[k]
            used[cand[k]] = True
    return float((idx >= 0).mean()) if len(idx) else 1.0, idx


# ------------------------------------------------------------------ key sources for the product's sampler
from probabilisticteacher_amd.modeling.sampling import keyed_perm_source, perm_key_source  # noqa: E402,F401
