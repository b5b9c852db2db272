"""CPU: the C restatement of Pillow's pixel arithmetic (oracle/csrc/ref_aug.c -- what the HIP augmentation kernels are
compared with on the GPU box) is PINNED against (a) tests/golden/augment.npz = outputs of the reference's own
GaussianBlur / Solarize classes and of the PIL calls torchvision's ColorJitter / RandomGrayscale make, and (b) the live
Pillow of this image when importable: every 7th of all 2^24 colours for the colour ops, random images for contrast / blur.
Plus the host logic of the data package (parameter sampling, box-blur weights, aspect-ratio grouping)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import augment as A

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.npz")


def _eq(a, b, what):
    assert a.shape == b.shape and bool((a == b).all()), f"{what}: {int((a != b).sum())} of {a.numel()} bytes differ"


def test_c_restatement_reproduces_the_reference_fixture():
    z = np.load(GOLD)
    img = torch.from_numpy(z["image"])
    for tag in ("a", "b", "c"):
        _eq(A.c_blur(img, float(z[f"blur_{tag}_sigma"][0])), torch.from_numpy(z[f"blur_{tag}"]), f"GaussianBlur {tag}")
    assert int(z["solarize_threshold"][0]) == 128
    _eq(A.c_solarize(img, 128), torch.from_numpy(z["solarize"]), "Solarize")
    for name, fn in (("brightness", A.c_brightness), ("contrast", A.c_contrast), ("saturation", A.c_saturation), ("hue", A.c_hue)):
        for j in range(2):
            _eq(fn(img, float(z[f"{name}_{j}_factor"][0])), torch.from_numpy(z[f"{name}_{j}"]), f"{name} {j}")
    _eq(A.c_gray(img), torch.from_numpy(z["gray"]), "grayscale")
    from probabilisticteacher_amd.data.augment import OP_BRIGHTNESS, OP_CONTRAST, OP_HUE, OP_SATURATION, StrongParams
    sat, hue, bri, con, sigma, thr = [float(v) for v in z["chain_params"]]
    p = StrongParams(jitter=[(OP_SATURATION, sat), (OP_HUE, hue), (OP_BRIGHTNESS, bri), (OP_CONTRAST, con)], gray=False,
                     blur_sigma=sigma, solarize=int(thr))
    _eq(A.apply_strong(img, p), torch.from_numpy(z["chain"]), "whole chain")


def test_c_restatement_equals_live_pillow():
    pytest.importorskip("PIL")
    v = np.arange(0, 1 << 24, 7, dtype=np.uint32)
    v = v[: (len(v) // 1024) * 1024]
    img = torch.from_numpy(np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255]).astype(np.uint8).reshape(3, -1, 1024))
    _eq(A.c_gray(img), A.pil_gray(img), "gray")
    _eq(A.c_solarize(img), A.pil_solarize(img), "solarize")
    for f in (0.6, 0.977, 1.0, 1.4):
        _eq(A.c_brightness(img, f), A.pil_brightness(img, f), f"brightness {f}")
        _eq(A.c_saturation(img, f), A.pil_saturation(img, f), f"saturation {f}")
    for hf in (-0.1, -0.031, 0.0, 0.1):
        _eq(A.c_hue(img, hf), A.pil_hue(img, hf), f"hue {hf}")
    rng = np.random.RandomState(0)
    noise = torch.from_numpy(rng.randint(0, 256, (3, 97, 131)).astype(np.uint8))
    smooth = torch.from_numpy((np.cumsum(rng.randint(-3, 4, (3, 97, 131)), axis=2) % 256).astype(np.uint8))
    for im in (noise, smooth):
        for f in (0.6, 1.25, 1.4):
            _eq(A.c_contrast(im, f), A.pil_contrast(im, f), f"contrast {f}")
        for s in (0.1, 0.35, 0.8, 1.234, 1.5, 1.9, 2.0):
            _eq(A.c_blur(im, s), A.pil_blur(im, s), f"blur {s}")


def test_parameter_sampling_weights_and_grouping():
    from probabilisticteacher_amd.data import AspectRatioGroupedSemiSupDatasetTwoCrop, sample_strong_params
    from probabilisticteacher_amd.data.augment import OP_HUE, box_blur_weights, hue_shift
    rng = random.Random(0)
    for _ in range(3000):                                     # python-side fixed-point weights == BoxBlur.c's (C restatement)
        s = rng.uniform(0.05, 3.0)
        assert box_blur_weights(s) == A.c_box_weights(s)
    assert hue_shift(-0.1) == (256 - 25) and hue_shift(0.1) == 25 and hue_shift(0.0) == 0 and A.hue_shift(-0.05) == hue_shift(-0.05)
    ps = [sample_strong_params(rng) for _ in range(4000)]
    frac = lambda f: sum(1 for p in ps if f(p)) / len(ps)
    assert abs(frac(lambda p: bool(p.jitter)) - 0.8) < 0.03 and abs(frac(lambda p: p.gray) - 0.2) < 0.03
    assert abs(frac(lambda p: p.blur_sigma is not None) - 0.5) < 0.03 and abs(frac(lambda p: p.solarize is not None) - 0.2) < 0.03
    for p in ps:
        assert len(p.jitter) in (0, 4) and (p.solarize in (None, 128)) and (p.blur_sigma is None or 0.1 <= p.blur_sigma <= 2.0)
        for op, f in p.jitter:
            assert (-0.1 <= f <= 0.1) if op == OP_HUE else (0.6 <= f <= 1.4)
    assert len({tuple(op for op, _ in p.jitter) for p in ps if p.jitter}) == 24          # all 4! orders occur
    # aspect-ratio grouping (pt/data/common.py:106-180): w > h and w <= h never share a batch
    def stream(sizes, tag):
        for i, (w, h) in enumerate(sizes):
            yield ({"width": w, "height": h, "id": (tag, i, "strong")}, {"width": w, "height": h, "id": (tag, i, "weak")})
    lab = [(8, 5), (5, 8), (9, 5), (5, 9), (7, 5), (5, 7), (8, 4), (4, 8)]
    unl = [(5, 8), (8, 5), (9, 4), (4, 9), (7, 5), (6, 5), (5, 7), (5, 6)]
    out = list(AspectRatioGroupedSemiSupDatasetTwoCrop((stream(lab, "l"), stream(unl, "u")), (2, 2)))
    assert len(out) >= 2
    for ls, lw, us, uw in out:
        assert len(ls) == len(lw) == len(us) == len(uw) == 2
        assert len({r["width"] > r["height"] for r in ls}) == 1 and len({r["width"] > r["height"] for r in us}) == 1
        assert [r["id"][:2] for r in ls] == [r["id"][:2] for r in lw] and all(r["id"][2] == "weak" for r in lw)


def test_resize_restatement_equals_live_pillow():
    """Image.resize(..., BILINEAR) of D2's ResizeTransform (the weak augmentation's ResizeShortestEdge): the C restatement
    of Pillow's Resample.c against the live Pillow -- down- and up-scaling, one-axis-only, Cityscapes -> 1333 x 666; and
    D2's size rule."""
    pytest.importorskip("PIL")
    from probabilisticteacher_amd.data import resize_shortest_edge_size
    rng = np.random.RandomState(1)
    for (H, W, nh, nw) in [(97, 131, 60, 81), (97, 131, 150, 200), (200, 400, 123, 247), (64, 64, 64, 100), (50, 70, 50, 70),
                           (1024, 2048, 666, 1333), (375, 500, 800, 1067), (31, 17, 9, 5), (600, 90, 40, 90)]:
        img = torch.from_numpy(rng.randint(0, 256, (3, H, W)).astype(np.uint8))
        _eq(A.c_resize(img, nh, nw), A.pil_resize(img, nh, nw), f"resize {H}x{W} -> {nh}x{nw}")
    assert resize_shortest_edge_size(1024, 2048, 600, 1200) == (600, 1200)
    assert resize_shortest_edge_size(1024, 2048, 800, 1333) == (667, 1333)        # long side capped: 1333 / 2048 * 1024 = 666.5
    assert resize_shortest_edge_size(375, 500, 800, 1333) == (800, 1067) and resize_shortest_edge_size(500, 375, 600, 1333) == (800, 600)
