"""GPU (pytest -m gpu): the device strong-augmentation kernels (csrc/augment.hip) against the CPU oracle's C restatement of
Pillow (oracle/csrc/ref_aug.c, itself pinned to the real Pillow and to the reference's classes in tests/test_augment_cpu.py)
and against the committed fixture of the reference's own GaussianBlur / Solarize outputs.  Byte outputs: BIT-EXACT."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import augment as A

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.npz")


def _eq(a, b, what):
    a, b = a.cpu(), b.cpu()
    assert a.shape == b.shape and bool((a == b).all()), f"{what}: {int((a != b).sum())} of {a.numel()} bytes differ"


def _img(rs, h, w, smooth=False):
    x = rs.randint(0, 256, (3, h, w))
    if smooth:
        x = np.cumsum(rs.randint(-3, 4, (3, h, w)), axis=2) % 256
    return torch.from_numpy(x.astype(np.uint8))


def test_device_ops_match_the_reference_fixture():
    from probabilisticteacher_amd.data.augment import (OP_BRIGHTNESS, OP_CONTRAST, OP_HUE, OP_SATURATION, StrongParams,
                                                       strong_augment_batch)
    z = np.load(GOLD)
    img = torch.from_numpy(z["image"]).to(DEV)
    cases = []
    for tag in ("a", "b", "c"):
        cases.append((f"blur_{tag}", StrongParams(blur_sigma=float(z[f"blur_{tag}_sigma"][0]))))
    cases.append(("solarize", StrongParams(solarize=int(z["solarize_threshold"][0]))))
    for name, op in (("brightness", OP_BRIGHTNESS), ("contrast", OP_CONTRAST), ("saturation", OP_SATURATION), ("hue", OP_HUE)):
        for j in range(2):
            cases.append((f"{name}_{j}", StrongParams(jitter=[(op, float(z[f"{name}_{j}_factor"][0]))])))
    cases.append(("gray", StrongParams(gray=True)))
    sat, hue, bri, con, sigma, thr = [float(v) for v in z["chain_params"]]
    cases.append(("chain", StrongParams(jitter=[(OP_SATURATION, sat), (OP_HUE, hue), (OP_BRIGHTNESS, bri), (OP_CONTRAST, con)],
                                        blur_sigma=sigma, solarize=int(thr))))
    # ONE batch: every image of it takes a different path through the rounds
    outs = strong_augment_batch([img] * len(cases), [p for _, p in cases])
    for (name, _), o in zip(cases, outs):
        _eq(o, torch.from_numpy(z[name]), name)
    assert torch.equal(img.cpu(), torch.from_numpy(z["image"])), "inputs are not modified"


def test_all_colours_and_random_draws_vs_oracle():
    from probabilisticteacher_amd.data.augment import (OP_BRIGHTNESS, OP_HUE, OP_SATURATION, StrongParams,
                                                       sample_strong_params, strong_augment_batch)
    # every 3rd of all 2^24 colours through the per-pixel ops
    v = np.arange(0, 1 << 24, 3, dtype=np.uint32)
    v = v[: (len(v) // 2048) * 2048]
    allc = torch.from_numpy(np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255]).astype(np.uint8).reshape(3, -1, 2048))
    ps = [StrongParams(jitter=[(OP_HUE, -0.1)]), StrongParams(jitter=[(OP_HUE, 0.0731)]), StrongParams(jitter=[(OP_SATURATION, 0.6)]),
          StrongParams(jitter=[(OP_SATURATION, 1.4)]), StrongParams(jitter=[(OP_BRIGHTNESS, 1.3999)]), StrongParams(gray=True),
          StrongParams(solarize=128)]
    outs = strong_augment_batch([allc.to(DEV)] * len(ps), ps)
    for p, o in zip(ps, outs):
        _eq(o, A.apply_strong(allc, p), f"all colours {p}")
    # random parameter draws on images of different sizes in one batch, incl. BASELINE size
    rs = np.random.RandomState(3)
    rng = random.Random(11)
    imgs = [_img(rs, 800, 1333), _img(rs, 800, 1333, smooth=True), _img(rs, 61, 47), _img(rs, 600, 800, smooth=True),
            _img(rs, 1, 9), _img(rs, 33, 1)] + [_img(rs, 96 + 8 * i, 128 - 4 * i, smooth=bool(i % 2)) for i in range(10)]
    params = [sample_strong_params(rng) for _ in imgs]
    params[0].blur_sigma, params[1].blur_sigma = 2.0, 0.1          # both box radii at full size
    params[0].jitter = params[0].jitter or [(OP_BRIGHTNESS, 0.9)]
    outs = strong_augment_batch([im.to(DEV) for im in imgs], params)
    for i, (im, p, o) in enumerate(zip(imgs, params, outs)):
        _eq(o, A.apply_strong(im, p), f"image {i} {tuple(im.shape)} {p}")


def test_hflip_and_two_crop_mapper_record_format():
    from probabilisticteacher_amd.data import DeviceTwoCropMapper, StrongParams, hflip_batch
    from probabilisticteacher_amd.structures import FreeInstances
    rs = np.random.RandomState(5)
    imgs = [_img(rs, 40, 67), _img(rs, 31, 32)]
    out = hflip_batch([im.to(DEV) for im in imgs], [True, False])
    _eq(out[0], imgs[0].flip(-1), "hflip")
    _eq(out[1], imgs[1], "no flip")
    dd = [{"image": imgs[0], "boxes": torch.tensor([[5.0, 4.0, 30.0, 20.0], [60.0, 1.0, 67.0, 39.0], [10.0, 10.0, 10.0, 30.0]]),
           "classes": torch.tensor([2, 5, 7]), "file_name": "a.png"},
          {"image": imgs[1], "file_name": "b.png"}]
    pairs = DeviceTwoCropMapper(DEV, seed=1)(dd, params=[StrongParams(solarize=128), StrongParams(gray=True)], flips=[True, False])
    (s0, w0), (s1, w1) = pairs
    _eq(w0["image"], imgs[0].flip(-1), "weak view = flipped image")
    _eq(s0["image"], A.c_solarize(imgs[0].flip(-1).contiguous()), "strong view = augmented weak view")
    assert s0["height"] == 40 and s0["width"] == 67 and s0["file_name"] == "a.png" and s0["image"].dtype == torch.uint8
    inst = s0["instances"]
    assert isinstance(inst, FreeInstances) and inst.image_size == (40, 67)
    # x -> w - x under the flip; the zero-width box is dropped (filter_empty_instances)
    assert torch.equal(inst.gt_boxes.tensor.cpu(), torch.tensor([[37.0, 4.0, 62.0, 20.0], [0.0, 1.0, 7.0, 39.0]]))
    assert inst.gt_classes.cpu().tolist() == [2, 5] and "instances" not in s1 and w1["image"].shape == (3, 31, 32)
    _eq(s1["image"], A.c_gray(imgs[1]), "second image")


def test_resize_and_full_weak_plus_strong_pipeline():
    """ResizeShortestEdge on the device (Pillow's antialiased bilinear resample, byte-exact with the C restatement that is
    pinned to the live Pillow) incl. Cityscapes -> 1333 x 667, up-scaling, one-axis-only; then the mapper end to end:
    resize -> flip -> strong augmentation, boxes scaled and flipped."""
    from probabilisticteacher_amd.data import DeviceTwoCropMapper, StrongParams, resize_batch
    rs = np.random.RandomState(8)
    cases = [(_img(rs, 97, 131), (60, 81)), (_img(rs, 97, 131, smooth=True), (150, 200)), (_img(rs, 64, 64), (64, 100)),
             (_img(rs, 50, 70), (50, 70)), (_img(rs, 1024, 2048, smooth=True), (667, 1333)), (_img(rs, 375, 500), (800, 1067)),
             (_img(rs, 600, 90), (40, 90)), (_img(rs, 31, 17), (9, 5))]
    outs = resize_batch([im.to(DEV) for im, _ in cases], [sz for _, sz in cases])
    for (im, (nh, nw)), o in zip(cases, outs):
        _eq(o, A.c_resize(im, nh, nw), f"resize {tuple(im.shape)} -> {(nh, nw)}")
    with pytest.raises(ValueError):
        resize_batch([cases[4][0].to(DEV)], [(30, 60)])                 # a 34x down-scale is outside the kernel's tap window
    img = _img(rs, 120, 200, smooth=True)
    dd = [{"image": img, "boxes": torch.tensor([[20.0, 30.0, 100.0, 90.0]]), "classes": torch.tensor([4])}]
    mp = DeviceTwoCropMapper(DEV, seed=0, min_size_train=(90,), max_size_train=120)
    (s, w), = mp(dd, params=[StrongParams(blur_sigma=1.3)], flips=[True])
    assert w["image"].shape == (3, 72, 120) and (w["height"], w["width"]) == (72, 120)     # 90 / 120 * 200 = 150 > 120 -> capped
    weak_ref = A.c_resize(img, 72, 120).flip(-1).contiguous()
    _eq(w["image"], weak_ref, "weak view = resized + flipped")
    _eq(s["image"], A.c_blur(weak_ref, 1.3), "strong view")
    want = torch.tensor([[120 - 100 * 0.6, 30 * 0.6, 120 - 20 * 0.6, 90 * 0.6]])
    assert torch.allclose(s["instances"].gt_boxes.tensor.cpu(), want, atol=1e-4)
