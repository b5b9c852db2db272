"""GPU parity at BASELINE size (pytest -m gpu): the shapes `bench.py` times, against stock torch CPU fp32 autograd
(conv stack, per layer group) and against the CPU oracle's full `run_step` (1333 x 800 images, final_c2f.yaml).

  * every distinct (Cin, Cout, H, W) of the VGG16 + RPN 3x3 stack at n = 2: forward, dgrad (incl. the fused ReLU-mask
    epilogue), wgrad and bias gradient through the production autograd nodes (`ops.vgg_block`, fused conv+ReLU+pool
    for the frozen blocks);
  * ONE n = 48 launch of conv3_2 (256 -> 256 at 200 x 333: the 8-wave variant, the n = 48 split-K count, the merged
    main + right-edge wgrad launch, epilogue 3) -- the launch shape of the joint student pass in the bench;
  * BASELINE configs[1]-shaped supervised student step (2 images) and configs[2]-shaped full mutual-learning step
    (1 labelled + 1 unlabelled image: EMA, teacher, pseudo labels, joint student pass, clip + SGD) vs the oracle.

Tolerances: conv outputs / gradients 1e-4 of the tensor's largest magnitude (fp32 MFMA k-order vs oneDNN blocking);
losses 1e-4 relative (BASELINE.json north_star); updated parameters 1e-4."""
import math
import random
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import d2, pt as opt
from tests.helpers import close, keyed_perm_source, match_detections

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MAX_TIED_ROWS_PERMUTED = 4      # per image and proposal stage: rows allowed to move inside a 1e-6-relative score tie (see _ProposalLog.check)


def _threads():
    torch.set_num_threads(max(2, min(os.cpu_count() or 2, 64)))


def _rel(a, b, tol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, f"{what}: {tuple(a.shape)} vs {tuple(b.shape)}"
    scale = float(b.abs().max())
    err = float((a - b).abs().max())
    assert err <= tol * scale + 1e-30, f"{what}: max abs err {err:.3e} vs {tol:.0e} x max|ref| {scale:.3e}"


def _block_backward_ref(acts, params, pool, gy, need_dx):
    """torch CPU backward of k x [conv3x3 + bias + ReLU] (+ 2x2 max pool) LINEARISED AT GIVEN ACTIVATIONS `acts` =
    [x, y1, .., yk]: ReLU / max-pool masks are taken from them.  The HIP backward is compared on identical masks: an
    fp32 pre-activation within rounding of zero has a different sign on the two sides (about one per million), which
    flips a whole 3 x 3 x C neighbourhood of the gradient -- a property of ReLU, not of either implementation (measured:
    stock torch fp32 vs fp64 shows the same ~1e-2 outliers at these sizes, tools/diag_fullsize.py)."""
    k = len(params) // 2
    d = gy
    if pool:
        yk = acts[k].clone().requires_grad_()
        F.max_pool2d(yk, 2, 2).backward(d)
        d = yk.grad
    d = d * (acts[k] > 0)
    grads = [None] * (2 * k)
    for j in range(k, 0, -1):
        xin = acts[j - 1].clone().requires_grad_(j > 1 or need_dx)
        w, b = params[2 * (j - 1)].clone().requires_grad_(), params[2 * (j - 1) + 1].clone().requires_grad_()
        F.conv2d(xin, w, b, padding=1).backward(d)
        grads[2 * (j - 1)], grads[2 * (j - 1) + 1] = w.grad, b.grad
        if j > 1:
            d = xin.grad * (acts[j - 1] > 0)
        elif need_dx:
            d = xin.grad
    return (d if need_dx else None), grads


# (name, cin, couts, H, W, pool, the block's input needs a gradient)
TRAINABLE_BLOCKS = [
    ("block3", 128, [256, 256, 256], 200, 333, True, False),     # conv3_1 (128->256), conv3_2/3 (256->256), 8-wave variant
    ("block4", 256, [512, 512, 512], 100, 166, True, True),      # conv4_1 (256->512), conv4_2/3 (512->512)
    ("block5", 512, [512, 512, 512], 50, 83, False, True),       # conv5_x (512->512), no pool (vgg.py:59)
]


@pytest.mark.parametrize("name,cin,couts,h,w,pool,need_dx", TRAINABLE_BLOCKS)
def test_trainable_vgg_blocks_full_size_vs_torch(name, cin, couts, h, w, pool, need_dx):
    from probabilisticteacher_amd import ops
    _threads()
    gen = torch.Generator().manual_seed(h + cin)
    n = 2
    x = torch.relu(torch.randn(n, cin, h, w, generator=gen))
    params, c = [], cin
    for co in couts:
        params += [torch.randn(co, c, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * c)), torch.randn(co, generator=gen) * 0.1]
        c = co
    # forward: every layer against stock torch (continuous in its inputs: no mask ambiguity)
    with torch.no_grad():
        yr = x
        for j in range(len(couts)):
            yr = F.relu(F.conv2d(yr, params[2 * j], params[2 * j + 1], padding=1))
        if pool:
            yr = F.max_pool2d(yr, 2, 2)
    xd = x.to(DEV).requires_grad_(need_dx)
    pd = [p.to(DEV).requires_grad_() for p in params]
    yd = ops.vgg_block(xd, pool, pd)
    _rel(yd, yr, 1e-4, f"{name} forward")
    gy = torch.randn(yr.shape, generator=gen)
    yd.backward(gy.to(DEV))
    # the block's own intermediate activations (same kernels, bitwise reproducible) define the masks of the reference
    with torch.no_grad():
        acts = [x]
        t = x.to(DEV)
        for j in range(len(couts)):
            t = ops.conv3x3(t, pd[2 * j].detach(), pd[2 * j + 1].detach(), True)
            acts.append(t.cpu())
    dx_ref, grads = _block_backward_ref(acts, params, pool, gy, need_dx)
    if need_dx:
        _rel(xd.grad, dx_ref, 1e-4, f"{name} dgrad")
    for j, (a, b) in enumerate(zip(pd, grads)):
        _rel(a.grad, b, 1e-4, f"{name} {'dW' if j % 2 == 0 else 'db'} of conv{j // 2 + 1}")


def _dilate3(m):
    return F.max_pool2d(m.float(), 3, 1, 1) > 0


@pytest.mark.parametrize("name,cin,couts,h,w,pool,need_dx", TRAINABLE_BLOCKS)
def test_trainable_vgg_blocks_input_gradient_elementwise_on_own_masks(name, cin, couts, h, w, pool, need_dx):
    """The block's input gradient against stock torch autograd running on ITS OWN ReLU / max-pool masks (no linearisation at
    the HIP activations), ELEMENTWISE: |a - b| <= 1e-4 |b| + 1e-5 max|b| at every pixel outside the region a mask decision can
    reach.  A mask decision is excluded only where it is undecidable in fp32 -- a pre-activation within 1e-6 of zero (relative
    to the layer's largest), a pool window whose two largest values are that close, or an element where the two sides' signs
    actually differ (counted and reported; about one per million) -- and its footprint is the 3 x 3 dilation per conv layer
    between it and the block input."""
    from probabilisticteacher_amd import ops
    _threads()
    gen = torch.Generator().manual_seed(h + cin + 1)
    n = 2
    x = torch.relu(torch.randn(n, cin, h, w, generator=gen))
    params, c = [], cin
    for co in couts:
        params += [torch.randn(co, c, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * c)), torch.randn(co, generator=gen) * 0.1]
        c = co
    k = len(couts)
    xr = x.clone().requires_grad_()
    zs, t = [], xr
    for j in range(k):
        z = F.conv2d(t, params[2 * j], params[2 * j + 1], padding=1)
        zs.append(z.detach())
        t = F.relu(z)
    yr = F.max_pool2d(t, 2, 2) if pool else t
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xd = x.to(DEV).requires_grad_()
    pd = [p.to(DEV) for p in params]
    yd = ops.vgg_block(xd, pool, pd)
    yd.backward(gy.to(DEV))
    with torch.no_grad():
        acts, a = [], x.to(DEV)
        for j in range(k):
            a = ops.conv3x3(a, pd[2 * j], pd[2 * j + 1], True)
            acts.append(a.cpu())
    # undecidable mask decisions, per layer, as pixel masks
    flips, reach = 0, torch.zeros(n, 1, h, w, dtype=torch.bool)
    for j in range(k - 1, -1, -1):
        z = zs[j]
        flipped = (acts[j] > 0) != (z > 0)
        flips += int(flipped.sum())
        risky = flipped | (z.abs() <= 1e-6 * float(z.abs().max()))
        if pool and j == k - 1:           # max-pool argmax ties
            y = F.relu(z)[:, :, : h // 2 * 2, : w // 2 * 2]
            win = y.unfold(2, 2, 2).unfold(3, 2, 2).reshape(n, y.shape[1], h // 2, w // 2, 4)
            top2 = win.topk(2, dim=-1).values
            tie = ((top2[..., 0] - top2[..., 1]) <= 1e-6 * float(y.max())) & (top2[..., 0] > 0)
            tie_px = tie.any(dim=1, keepdim=True).float()
            tie_full = F.interpolate(tie_px, scale_factor=2, mode="nearest") > 0
            risky_px = risky.any(dim=1, keepdim=True)
            risky_px[:, :, : h // 2 * 2, : w // 2 * 2] |= tie_full
        else:
            risky_px = risky.any(dim=1, keepdim=True)
        reach = _dilate3(reach | risky_px)           # through conv j's dgrad
    keep = ~reach.expand_as(xr.grad)
    frac = float(keep.float().mean())
    assert frac > 0.5, f"{name}: only {frac:.2f} of the pixels are outside the reach of an undecidable mask decision"
    got, ref = xd.grad.cpu(), xr.grad
    err = (got - ref).abs()
    tol = 1e-4 * ref.abs() + 1e-5 * float(ref.abs().max())
    bad = (err > tol) & keep
    assert not bool(bad.any()), (f"{name}: {int(bad.sum())} input-gradient elements off outside the excluded region "
                                 f"(worst {float(err[keep].max()):.3e}, max|ref| {float(ref.abs().max()):.3e})")
    print(f"\n[{name}] dgrad elementwise on own masks: {frac:.3f} of the pixels compared, {flips} sign flips of "
          f"{sum(z.numel() for z in zs)} pre-activations, worst err {float(err[keep].max()):.2e} (max|ref| {float(ref.abs().max()):.2e})")


def test_frozen_blocks_and_rpn_conv_full_size_vs_torch():
    """blocks 1-2 (frozen: stem kernel, fused conv + ReLU + pool epilogue) at 800 x 1333 / 400 x 666 and the RPN 3x3
    conv (512 -> 512 + ReLU on the 50 x 83 map, all three gradients)"""
    from probabilisticteacher_amd import ops
    _threads()
    gen = torch.Generator().manual_seed(1)
    n = 2
    x = torch.randn(n, 3, 800, 1333, generator=gen)
    ws = []
    for ci, co in ((3, 64), (64, 64), (64, 128), (128, 128)):
        ws.append((torch.randn(co, ci, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * ci)), torch.randn(co, generator=gen) * 0.1))
    with torch.no_grad():
        r = F.relu(F.conv2d(x, *ws[0], padding=1))
        d = ops.conv3x3(x.to(DEV), ws[0][0].to(DEV), ws[0][1].to(DEV), True)
        _rel(d, r, 1e-4, "conv1_1 (stem)")
        r = F.max_pool2d(F.relu(F.conv2d(r, *ws[1], padding=1)), 2, 2)
        d = ops.conv3x3_relu_pool_nograd(d, ws[1][0].to(DEV), ws[1][1].to(DEV))
        _rel(d, r, 1e-4, "conv1_2 + pool")
        r = F.relu(F.conv2d(r, *ws[2], padding=1))
        d = ops.conv3x3(d, ws[2][0].to(DEV), ws[2][1].to(DEV), True)
        _rel(d, r, 1e-4, "conv2_1")
        r = F.max_pool2d(F.relu(F.conv2d(r, *ws[3], padding=1)), 2, 2)
        d = ops.conv3x3_relu_pool_nograd(d, ws[3][0].to(DEV), ws[3][1].to(DEV))
        _rel(d, r, 1e-4, "conv2_2 + pool")
        assert d.shape == (n, 128, 200, 333)
    f = torch.relu(torch.randn(n, 512, 50, 83, generator=gen))
    wt, b = torch.randn(512, 512, 3, 3, generator=gen) * 0.01, torch.zeros(512)
    with torch.no_grad():
        yr = F.relu(F.conv2d(f, wt, b, padding=1))
    fd, wd, bd = (t.to(DEV).requires_grad_() for t in (f, wt, b))
    yd = ops.conv3x3(fd, wd, bd, True)
    _rel(yd, yr, 1e-4, "rpn conv forward")
    gy = torch.randn(yr.shape, generator=gen)
    yd.backward(gy.to(DEV))
    dx_ref, (dw_ref, db_ref) = _block_backward_ref([f, yd.detach().cpu()], [wt, b], False, gy, True)
    _rel(fd.grad, dx_ref, 1e-4, "rpn conv dgrad")
    _rel(wd.grad, dw_ref, 1e-4, "rpn conv dW")
    _rel(bd.grad, db_ref, 1e-4, "rpn conv db")


def test_conv3_2_at_n48_bench_launch_shape_vs_torch():
    """ONE launch of each conv3_2 kernel at the joint student pass's batch (n = 48 = 32 labelled views + 16 unlabelled):
    forward (epilogue 1), dgrad with the producer's ReLU mask (epilogue 3) on the F(4x4,3x3) kernel (csrc/wino4.hip); weight +
    bias gradient on the F(4x4,3x3)-domain kernel (csrc/wino4w.hip, the one the fp32 step runs) AND on the direct split-K
    kernel (n = 48 split count, merged main + right-edge launch).  The routes are asserted, not assumed."""
    from probabilisticteacher_amd import ops
    from probabilisticteacher_amd import _lib
    _threads()
    gen = torch.Generator().manual_seed(48)
    n, c, h, w = 48, 256, 200, 333
    x = torch.relu(torch.randn(n, c, h, w, generator=gen))          # post-ReLU activation of conv3_1 (has zeros)
    wt = torch.randn(c, c, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * c))
    b = torch.randn(c, generator=gen) * 0.1
    gy = torch.randn(n, c, h, w, generator=gen)
    xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.conv2d(xr, wr, br, padding=1)
    yrelu = F.relu(yr)
    yr.backward(gy)
    xd, wd, bd, gd = x.to(DEV), wt.to(DEV), b.to(DEV), gy.to(DEV)
    assert ops._conv_kind(c, c, (h, w)) == "wino4", "conv3_2 forward / dgrad must route to the F(4x4,3x3) kernel"
    yd = ops.conv3x3_raw(xd, ops.conv3x3_pack(wd, 0, 1, (h, w)), bd, None, c, 1)
    _rel(yd, yrelu.detach(), 1e-4, "conv3_2 forward n=48")
    del yd
    dx = ops.conv3x3_raw(gd, ops.conv3x3_pack(wd, 1, 3, (h, w)), None, xd, c, 3)
    _rel(dx, xr.grad * (x > 0), 1e-4, "conv3_2 dgrad + mask n=48")
    del dx
    # the kernel the fp32 step runs (ops.conv3x3_wgrad routes this shape to ptmi_conv3x3_wino4_wgrad): 3.3 GB per operand -- the
    # shape where its 32-bit buffer offsets, split count and chunk clamp matter
    assert ops._wgrad_kind(c, c, h, w) == "wino4w", "conv3_2 weight gradient must route to the F(4x4,3x3)-domain kernel"
    ops.profile_start()
    dw, db = ops.conv3x3_wgrad(xd, gd, c)
    assert list(ops.profile_stop()) == ["conv3x3_wino4_wgrad"]
    _rel(dw, wr.grad, 1e-4, "conv3_2 Winograd wgrad n=48")
    _rel(db, br.grad, 1e-4, "conv3_2 Winograd bias grad n=48")
    dw2, db2 = ops.conv3x3_wgrad(xd, gd, c)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "the split reduction runs in a fixed order"
    # ... and the direct split-K kernel (bf16 path, small channel counts) at the same shape
    dw, db = torch.empty_like(wd), torch.empty(c, device=DEV)
    nws = _lib.load().ptmi_conv3x3_wgrad_ws_floats(n, c, c, h, w)
    ws = torch.empty(nws, device=DEV)
    _lib.call("ptmi_conv3x3_wgrad", ops._ptr(xd), ops._ptr(gd), ops._ptr(dw), ops._ptr(db), ops._ptr(ws), n, c, c, h, w, 0,
              ops._stream())
    _rel(dw, wr.grad, 1e-4, "conv3_2 direct wgrad n=48")
    _rel(db, br.grad, 1e-4, "conv3_2 direct bias grad n=48")


def test_conv4_2_at_n48_bench_launch_shape_vs_torch():
    """The 512-channel launch shape of the joint student pass (n = 48, 512 -> 512 at 100 x 166): F(4x4,3x3) forward (8 channel
    tiles), dgrad with the producer's ReLU mask, F(4x4,3x3)-domain weight + bias gradient (chunks of 4 rows x 16 columns, fill
    0.94), each as ONE launch against torch CPU fp32; routes asserted."""
    from probabilisticteacher_amd import ops
    from probabilisticteacher_amd import _lib
    _threads()
    gen = torch.Generator().manual_seed(4802)
    n, c, h, w = 48, 512, 100, 166
    x = torch.relu(torch.randn(n, c, h, w, generator=gen))
    wt = torch.randn(c, c, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * c))
    b = torch.randn(c, generator=gen) * 0.1
    gy = torch.randn(n, c, h, w, generator=gen)
    xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.conv2d(xr, wr, br, padding=1)
    yrelu = F.relu(yr)
    yr.backward(gy)
    xd, wd, bd, gd = x.to(DEV), wt.to(DEV), b.to(DEV), gy.to(DEV)
    assert ops._conv_kind(c, c, (h, w)) == "wino4" and ops._wgrad_kind(c, c, h, w) == "wino4w"
    yd = ops.conv3x3_raw(xd, ops.conv3x3_pack(wd, 0, 1, (h, w)), bd, None, c, 1)
    _rel(yd, yrelu.detach(), 1e-4, "conv4_2 forward n=48")
    del yd
    dx = ops.conv3x3_raw(gd, ops.conv3x3_pack(wd, 1, 3, (h, w)), None, xd, c, 3)
    _rel(dx, xr.grad * (x > 0), 1e-4, "conv4_2 dgrad + mask n=48")
    del dx
    ops.profile_start()
    dw, db = ops.conv3x3_wgrad(xd, gd, c)
    assert list(ops.profile_stop()) == ["conv3x3_wino4_wgrad"]
    _rel(dw, wr.grad, 1e-4, "conv4_2 Winograd wgrad n=48")
    _rel(db, br.grad, 1e-4, "conv4_2 Winograd bias grad n=48")


def test_conv5_2_at_n48_bench_launch_shape_vs_torch():
    """The 50 x 83 launch shape of the joint student pass (n = 48, 512 -> 512; conv5_x and the RPN conv): F(4x4,3x3) forward and
    dgrad, and the weight + bias gradient on whichever kernel ops.conv3x3_wgrad routes this map to (the route is printed and its
    profile key asserted: round 5 kept it on the F(2x2,3x3)-domain kernel conv3x3_wino_wgrad_kernel<7>, chunk fill 0.83), each
    as ONE launch against torch CPU fp32.  Both weight-gradient kernels are checked at this shape, whatever the route."""
    from probabilisticteacher_amd import ops
    from probabilisticteacher_amd import _lib
    _threads()
    gen = torch.Generator().manual_seed(4852)
    n, c, h, w = 48, 512, 50, 83
    x = torch.relu(torch.randn(n, c, h, w, generator=gen))
    wt = torch.randn(c, c, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * c))
    b = torch.randn(c, generator=gen) * 0.1
    gy = torch.randn(n, c, h, w, generator=gen)
    xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.conv2d(xr, wr, br, padding=1)
    yrelu = F.relu(yr)
    yr.backward(gy)
    xd, wd, bd, gd = x.to(DEV), wt.to(DEV), b.to(DEV), gy.to(DEV)
    assert ops._conv_kind(c, c, (h, w)) == "wino4"
    yd = ops.conv3x3_raw(xd, ops.conv3x3_pack(wd, 0, 1, (h, w)), bd, None, c, 1)
    _rel(yd, yrelu.detach(), 1e-4, "conv5_2 forward n=48")
    del yd
    dx = ops.conv3x3_raw(gd, ops.conv3x3_pack(wd, 1, 3, (h, w)), None, xd, c, 3)
    _rel(dx, xr.grad * (x > 0), 1e-4, "conv5_2 dgrad + mask n=48")
    del dx
    kind = ops._wgrad_kind(c, c, h, w)
    print(f"conv5_2 n=48 weight-gradient route: {kind}")
    assert kind in ("wino", "wino4w")
    ops.profile_start()
    dw, db = ops.conv3x3_wgrad(xd, gd, c)
    assert list(ops.profile_stop()) == [{"wino": "conv3x3_wino_wgrad", "wino4w": "conv3x3_wino4_wgrad"}[kind]]
    _rel(dw, wr.grad, 1e-4, f"conv5_2 routed ({kind}) wgrad n=48")
    _rel(db, br.grad, 1e-4, f"conv5_2 routed ({kind}) bias grad n=48")
    lib = _lib.load()
    for name in ("ptmi_conv3x3_wino_wgrad", "ptmi_conv3x3_wino4_wgrad"):       # both Winograd-domain kernels, called by name
        assert getattr(lib, name + "_fits")(h, w)
        dw, db = torch.empty_like(wd), torch.empty(c, device=DEV)
        ws = torch.empty(getattr(lib, name + "_ws_floats")(n, c, c, h, w), device=DEV)
        _lib.call(name, ops._ptr(xd), ops._ptr(gd), ops._ptr(dw), ops._ptr(db), ops._ptr(ws), n, c, c, h, w, 0, ops._stream())
        _rel(dw, wr.grad, 1e-4, f"conv5_2 {name} n=48")
        _rel(db, br.grad, 1e-4, f"conv5_2 {name} bias grad n=48")


def _records(gen, n_img, h, w, K, m0=3):
    from probabilisticteacher_amd.structures import Boxes, FreeInstances
    recs, orecs = [], []
    for i in range(n_img):
        img = torch.randint(0, 256, (3, h, w), generator=gen, dtype=torch.uint8)
        m = m0 + i
        xy = torch.rand(m, 2, generator=gen) * torch.tensor([w * 0.6, h * 0.6])
        boxes = torch.cat([xy, xy + 40 + torch.rand(m, 2, generator=gen) * 300], 1)
        boxes[:, 0::2].clamp_(0, w)
        boxes[:, 1::2].clamp_(0, h)
        cls = torch.randint(0, K, (m,), generator=gen)
        a, b = FreeInstances((h, w)), opt.FreeInstances((h, w))
        a.gt_boxes, a.gt_classes = Boxes(boxes.clone()), cls.clone()
        b.gt_boxes, b.gt_classes = d2.Boxes(boxes.clone()), cls.clone()
        recs.append({"image": img, "height": h, "width": w, "instances": a})
        orecs.append({"image": img, "height": h, "width": w, "instances": b})
    return recs, orecs


PROBES = ("roi_heads.box_predictor.cls_score.weight", "proposal_generator.rpn_head.conv.bias",
          "backbone.vgg_block5.0.conv3.weight", "backbone.vgg_block3.0.conv1.weight", "roi_heads.box_head.fc1.bias")


def _load(model, params):
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in params.items():
            sd[k].copy_(v)


def _spread(params, teacher=False):
    """Random-init heads emit near-tied scores (objectness logits ~ 1e-1, class probabilities ~ 1/9): ranks, NMS survivors
    and with them the position-indexed ROI sample would hinge on the last ulp.  Scale the score heads so that the scores
    are spread like a trained model's (identical parameters on both sides, so parity is unaffected)."""
    p = {k: v.clone() for k, v in params.items()}
    p["proposal_generator.rpn_head.objectness_logits.weight"] *= 30.0
    if teacher:
        p["roi_heads.box_predictor.cls_score.weight"] *= 3.0
        p["roi_heads.box_predictor.cls_score.bias"][-1] += 2.0          # background-heavy: few candidates above 0.05
    return p


class _ProposalLog:
    """The proposal stage is the one place where the two sides cannot agree END TO END at this size by exact equality: 12 000
    fp32 scores per image are denser than the accumulated rounding differences of a 14-layer conv stack (~1e-5 relative), so
    adjacent ranks swap, and a single swap re-orders the proposal list and with it the position-indexed ROI sample -- on any
    two fp32 implementations (cuDNN vs oneDNN as much as MFMA vs oneDNN).  So the end-to-end statement is made in two
    deterministic halves, per call of the stage and per image (`check`):
      (i)  the stage's INPUTS on the two sides -- objectness logits, decoded boxes, sigma logits of all 37 350 anchors --
           agree elementwise: |a - b| <= 1e-4 |b| + 1e-5 max|b|;
      (ii) the HIP stage's OUTPUT equals the ORACLE's `find_top_rpn_proposals` run on the HIP side's inputs EXACTLY: same
           count, the same boxes bit for bit in the same order (scores 1e-5: expf).  Every difference between the two sides'
           proposal lists is therefore a consequence of (i)-sized input noise passing through exact, identical decision logic
           -- there is no room left for a ranking or NMS defect of the HIP stage, however small a fraction it would touch.
    The oracle's own proposals are still computed and reported against the HIP ones (rows in common, first differing row).
    For everything AFTER this stage the oracle's calls are answered with the HIP proposals (in call order), so that the
    later stages are compared on identical proposals."""

    def __init__(self, monkeypatch, ocfg=None):
        from probabilisticteacher_amd.modeling import rpn as hip_rpn
        self.hip, self.ref, self.calls, self.hip_in, self.ref_in, self.ocfg = [], [], [], [], [], ocfg
        f_hip, f_ref = hip_rpn.find_top_rpn_proposals, opt.find_top_rpn_proposals

        # records are kept PER IMAGE: the two sides may group the same images into different calls of the stage (the HIP joint
        # student pass predicts the proposals of both branches in one call, the oracle one call per branch)
        def w_hip(decoded, logits, sigma_logits, image_sizes, nms_thresh, pre_nms_topk, post_nms_topk, min_box_size, training):
            out = f_hip(decoded, logits, sigma_logits, image_sizes, nms_thresh, pre_nms_topk, post_nms_topk, min_box_size,
                        training)
            n = len(out)
            dec, lg, sg = decoded.detach().cpu().view(n, -1, 4), logits.detach().cpu().view(n, -1), sigma_logits.detach().cpu().view(n, -1, 4)
            for i, o in enumerate(out):
                rec = (o.proposal_boxes.tensor.cpu(), o.objectness_logits.cpu(), o.image_size)
                self.calls.append(rec)
                self.hip.append(rec[0])
                self.hip_in.append((dec[i].clone(), lg[i].clone(), sg[i].clone(), image_sizes[i], pre_nms_topk, post_nms_topk, training,
                                    rec[:2]))
            return out

        def w_ref(cfg, proposals, logits, image_sizes, sigma_logits, pre_nms_topk, post_nms_topk, training):
            own = f_ref(cfg, proposals, logits, image_sizes, sigma_logits, pre_nms_topk, post_nms_topk, training)
            n = len(own)
            self.ref += [o.proposal_boxes.tensor.clone() for o in own]
            pr, lg, sg = proposals.detach().view(n, -1, 4), logits.detach().view(n, -1), sigma_logits.detach().view(n, -1, 4)
            self.ref_in += [(pr[i].clone(), lg[i].clone(), sg[i].clone()) for i in range(n)]
            assert len(self.calls) >= n, "the two sides send the images through the proposal stage in the same order"
            out = []
            for _ in range(n):
                boxes, lgt, size = self.calls.pop(0)
                r = opt.FreeInstances(size)
                r.proposal_boxes, r.objectness_logits = d2.Boxes(boxes.clone()), lgt.clone()
                out.append(r)
            return out
        self._f_ref = f_ref
        monkeypatch.setattr(hip_rpn, "find_top_rpn_proposals", w_hip)
        monkeypatch.setattr(opt, "find_top_rpn_proposals", w_ref)

    def check(self, ocfg=None):
        """(i) + (ii) of the class comment for every image that went through the stage; returns a one-line report"""
        ocfg = ocfg or self.ocfg
        assert len(self.hip) == len(self.ref) and not self.calls and len(self.hip_in) == len(self.ref_in)
        out = []
        for k, ((dec, lg, sg, size, pre, post, training, (hb, hs)), (odec, olg, osg)) in enumerate(zip(self.hip_in, self.ref_in)):
            for name, a, b in (("logits", lg, olg), ("decoded boxes", dec, odec), ("sigma logits", sg, osg)):
                err = (a.double() - b.double()).abs()
                tol = 1e-4 * b.double().abs() + 1e-5 * float(b.abs().max())
                assert bool((err <= tol).all()), (f"proposal-stage input {name}: {int((err > tol).sum())} of {err.numel()} elements "
                                                  f"beyond 1e-4 |b| + 1e-5 max|b| (worst {float(err.max()):.3e})")
            r = self._f_ref(ocfg, dec.unsqueeze(0), lg.unsqueeze(0), [size], sg.unsqueeze(0), pre, post, training)[0]
            rb, rs = r.proposal_boxes.tensor, r.objectness_logits
            assert len(hb) == len(rb), f"proposal count {len(hb)} vs the oracle's stage on the same inputs {len(rb)}"
            if not torch.equal(hb, rb):
                # The kept SET and its order must be the oracle's -- except inside runs of proposals whose sigma-rescored scores
                # (proposal_utils.py:136-138: logit x (1 - mean sigmoid(sigma logits))) are tied to within 1e-6 relative: the
                # rescoring goes through a sigmoid whose last ulps differ between torch's CPU kernel and the device's expf, so two
                # proposals 1 ulp apart may swap (first seen at 8 + 8 images: rows 1599 / 1600 of one image, scores 87.426567 /
                # 87.426559).  Inside such a run the rows are compared as a set; everything else bit for bit.
                srt = rs.double()
                run = torch.zeros(len(rs), dtype=torch.long)
                run[1:] = torch.cumsum(((srt[:-1] - srt[1:]).abs() > 1e-6 * srt[:-1].abs()).long(), 0)
                bad = (hb != rb).any(dim=1)
                n_perm = int(bad.sum())
                for rid in run[bad].unique().tolist():
                    rows = (run == rid).nonzero()[:, 0]
                    a = sorted(map(tuple, hb[rows].tolist()))
                    b = sorted(map(tuple, rb[rows].tolist()))
                    assert len(rows) > 1 and a == b, ("HIP proposals differ from the oracle's find_top_rpn_proposals run on the SAME "
                                                      f"inputs beyond a permutation of score-tied rows: rows {rows.tolist()}")
                # the relaxation is BOUNDED and reported: at most MAX_TIED_ROWS_PERMUTED rows per image may sit at another position
                # of their tie run (north_star: keep-masks bit-exact -- the kept SET is still exact, only the order inside a tie moves)
                print(f"  proposal stage call {k}: {n_perm} of {len(hb)} proposal rows permuted inside 1e-6-relative score ties "
                      f"(cap {MAX_TIED_ROWS_PERMUTED})")
                self.permuted_rows = getattr(self, "permuted_rows", 0) + n_perm
                assert n_perm <= MAX_TIED_ROWS_PERMUTED, (f"{n_perm} proposal rows of stage call {k} are permuted inside score ties: more "
                                                          f"than the {MAX_TIED_ROWS_PERMUTED} a last-ulp sigmoid difference explains")
                hs = hs.clone()
                hs_sorted, rs_sorted = torch.sort(hs, descending=True)[0], torch.sort(rs, descending=True)[0]
                close(hs_sorted, rs_sorted, 1e-5, 1e-6, "proposal scores on identical inputs (tied rows permuted)")
            else:
                close(hs, rs, 1e-5, 1e-6, "proposal scores on identical inputs")
            a, b = self.hip[k], self.ref[k]
            za, zb = np.zeros(len(a), np.int64), np.zeros(len(b), np.int64)
            frac, _ = match_detections(a, za, b, zb, box_tol=5e-3)
            n = min(len(a), len(b))
            rows = ((a[:n] - b[:n]).abs() <= 1e-2).all(dim=1)
            out.append(f"{len(a)}/{len(b)} boxes (exact vs the oracle stage on HIP inputs), {frac:.4f} in common with the "
                       f"oracle's own, first differing row {int((~rows).nonzero()[0]) if not bool(rows.all()) else -1}")
        return "; ".join(out)

    check_sets = check


def _compare_step(m, om, tr, state, params, keys, tag):
    """both sides saw the same proposals and the same sampler keys: every loss to 1e-4 (north_star), the gradient norm,
    the updated parameters"""
    for k in keys:
        assert math.isfinite(om[k]), f"{tag} {k}: the test inputs must keep every loss finite (oracle: {om[k]})"
        close(torch.tensor(m[k]), torch.tensor(om[k]), 1e-4, 1e-6, f"{tag} {k}")
    close(torch.tensor(m["total_loss"]), torch.tensor(om["total_loss"]), 1e-4, 1e-6, f"{tag} total_loss")
    close(torch.tensor(m["grad_norm"]), torch.tensor(om["grad_norm"]), 1e-3, 1e-6, f"{tag} grad_norm")
    sd = tr.model.state_dict()
    for k in PROBES:
        ref = state["student"][k].detach()
        assert not torch.equal(ref, params[k]), k + " must have been updated"
        close(sd[k].cpu(), ref, 1e-4, 1e-5 * float(ref.abs().max()) + 1e-7, f"{tag} updated {k}")


def test_baseline_config1_supervised_step_1333x800_vs_oracle(monkeypatch, capsys):
    """BASELINE.json configs[1] shape: final_c2f.yaml (K = 8), student-only supervised forward/backward + clip + SGD on
    1333 x 800 images (here the strong + weak view of one labelled image = a batch of 2; the bench runs 8)."""
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer
    from probabilisticteacher_amd.modeling import sampling
    _threads()
    cfg = setup_cfg("configs/pt/final_c2f.yaml", ["MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 10 ** 6,
                                                  "SOLVER.IMG_PER_BATCH_LABEL", 1, "SOLVER.IMG_PER_BATCH_UNLABEL", 1])
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    ocfg = opt.Cfg(num_classes=K, anchor_generator=cfg.MODEL.ANCHOR_GENERATOR.NAME, burn_up_step=10 ** 6)
    params = _spread(opt.golden_params(ocfg, 31))
    ratios = [0.9, 0.55]
    it = iter(list(ratios))
    tr = PTrainer(cfg, ratio_fn=lambda: next(it))
    _load(tr.model, params)
    _load(tr.model_teacher, params)
    recs, orecs = _records(torch.Generator().manual_seed(3), 2, 800, 1333, K)
    log = _ProposalLog(monkeypatch, ocfg)
    kp = opt.KeyedPerm(51)
    sampling.set_key_source(keyed_perm_source(kp))
    try:
        m = tr.run_step(([recs[0]], [recs[1]], [recs[0]], [recs[1]]))
    finally:
        sampling.set_key_source(None)
    kp.start_replay()
    state = {"student": {k: v.clone() for k, v in params.items()}, "teacher": {k: v.clone() for k, v in params.items()},
             "bufs": {}, "iter": 0}
    om = opt.run_step(ocfg, state, ([orecs[0]], [orecs[1]], [orecs[0]], [orecs[1]]), {"label": ratios, "unlabel": []},
                      perm_fn=kp)
    with capsys.disabled():
        print(f"\n[configs[1] 1333x800] proposals: {log.check_sets()}; losses HIP {m} oracle {om}")
    _compare_step(m, om, tr, state, params, ["loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg"], "configs[1]")


def _spread_k1(params):
    """K = 1 (final_s2c.yaml: one foreground class + background): spread the objectness scores as `_spread` does and make the
    class head foreground-leaning, so that the teacher's detections carry foreground soft labels -- the unsupervised box
    terms are means over the rows whose teacher argmax is foreground (fast_rcnn.py:215-263) and NaN when there are none."""
    p = {k: v.clone() for k, v in params.items()}
    p["proposal_generator.rpn_head.objectness_logits.weight"] *= 30.0
    p["roi_heads.box_predictor.cls_score.weight"] *= 3.0
    p["roi_heads.box_predictor.cls_score.bias"][0] += 1.0
    return p


def mutual_learning_step_vs_oracle(monkeypatch, yaml, h, w, n_img=1, seed=33, spread=None, extra_cfg=(), rounding=None,
                                   oracle=True, paired_views=False, ratio_range=(0.55, 0.95), pseudo_from=None):
    """One full mutual-learning PTrainer.run_step (BURN_UP_STEP = 0: EMA copy, teacher forward + pseudo labels, shrink-paste,
    joint supervised + unsupervised student pass, one backward, clip + SGD) on n_img labelled + n_img unlabelled h x w images,
    and the oracle's run_step on the same records, shrink ratios and sampler keys.  The oracle's proposal calls are answered
    with the HIP proposals and its student is handed the HIP teacher's pseudo labels (see _ProposalLog); the two teachers'
    pseudo labels are compared with each other.  paired_views: the strong (student) view of an unlabelled image is its weak
    (teacher) view plus pixel noise, as in the real two-crop pipeline -- the student's proposals then overlap the teacher's
    pseudo boxes, so the unsupervised ROI terms have rows to average over.  pseudo_from: pseudo labels of an earlier run to
    hand to this run's student instead of its own teacher's (comparisons between two HIP runs whose teachers would otherwise
    make different index decisions).  Returns everything the callers assert on."""
    from probabilisticteacher_amd import ops
    from probabilisticteacher_amd.config import setup_cfg
    from probabilisticteacher_amd.engine import PTrainer
    from probabilisticteacher_amd.modeling import sampling
    _threads()
    cfg = setup_cfg(yaml, ["MODEL.DEVICE", DEV, "MODEL.VGG.PRETRAIN", "", "UNSUPNET.BURN_UP_STEP", 0,
                           "SOLVER.IMG_PER_BATCH_LABEL", n_img, "SOLVER.IMG_PER_BATCH_UNLABEL", n_img, *extra_cfg])
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    ocfg = opt.Cfg(num_classes=K, anchor_generator=cfg.MODEL.ANCHOR_GENERATOR.NAME, burn_up_step=0,
                   tau=tuple(cfg.UNSUPNET.TAU))
    # EMA with keep_rate 0 at iter == BURN_UP_STEP copies the student into the teacher: the step's teacher IS the student
    spread = spread or (lambda p: _spread(p, teacher=True))
    params = spread(opt.golden_params(ocfg, seed))
    rr = random.Random(seed)
    r_unlabel = [rr.uniform(*ratio_range) for _ in range(n_img)]
    r_label = [rr.uniform(*ratio_range) for _ in range(n_img)]
    seq = iter(r_unlabel + r_label)                     # run_step resizes unlabel_q first, then label_q (trainer.py:329-330)

    class Recording(PTrainer):
        mine = None

        def process_pseudo_label(self, proposals, proposal_type, psedo_label_method=""):
            out, nn = super().process_pseudo_label(proposals, proposal_type, psedo_label_method)
            if pseudo_from is not None:
                out = pseudo_from
            self.mine = out
            return out, nn

    tr = Recording(cfg, ratio_fn=lambda: next(seq))
    if rounding is not None:
        tr.operand_rounding = rounding
    _load(tr.model, params)
    _load(tr.model_teacher, opt.golden_params(ocfg, seed + 1))      # overwritten by the EMA copy
    assert tr.joint_student_pass
    g = torch.Generator().manual_seed(seed + 2)
    lab, olab = _records(g, 2 * n_img, h, w, K)         # label_q, label_k
    unl, ounl = _records(g, 2 * n_img, h, w, K)         # unlabel_q, unlabel_k (their ground truth is dropped)
    if paired_views:
        for i in range(n_img):
            weak = unl[n_img + i]["image"]
            strong = (weak.int() + torch.randint(-12, 13, weak.shape, generator=g)).clamp(0, 255).to(torch.uint8)
            unl[i]["image"], ounl[i]["image"] = strong, strong.clone()
    log = _ProposalLog(monkeypatch, ocfg)
    kp = opt.KeyedPerm(seed + 28)
    sampling.set_key_source(keyed_perm_source(kp))
    try:
        m = tr.run_step((lab[:n_img], lab[n_img:], unl[:n_img], unl[n_img:]))
    finally:
        sampling.set_key_source(None)
    out = {"m": m, "tr": tr, "params": params, "log": log, "K": K}
    if not oracle:
        return out
    kp.start_replay()
    override = []
    for p in tr.mine:
        o = opt.FreeInstances(p.image_size)
        o.pseudo_boxes = d2.Boxes(p.pseudo_boxes.tensor.cpu().clone())
        o.scores_logists, o.boxes_sigma = p.scores_logists.cpu().clone(), p.boxes_sigma.cpu().clone()
        override.append(o)
    state = {"student": {k: v.clone() for k, v in params.items()},
             "teacher": {k: v.clone() for k, v in opt.golden_params(ocfg, seed + 1).items()}, "bufs": {}, "iter": 0}
    om = opt.run_step(ocfg, state, (olab[:n_img], olab[n_img:], ounl[:n_img], ounl[n_img:]),
                      {"label": r_label, "unlabel": r_unlabel}, perm_fn=kp, pseudo_override=override)
    out.update(om=om, state=state)
    return out


def check_teacher_and_pseudo_labels(res, logit_tol=(1e-3, 5e-4)):
    """EMA copy exact; the two teachers' pseudo labels (same proposals in) are the same boxes, order-insensitive"""
    tr, state, params = res["tr"], res["state"], res["params"]
    tsd = tr.model_teacher.state_dict()
    for k in PROBES:
        assert torch.equal(tsd[k].cpu(), params[k]) and torch.equal(state["teacher"][k], params[k]), "EMA copy " + k
    for mine, ref in zip(tr.mine, state["last_pseudo"]):
        assert 0 < len(ref) <= 100 and len(mine) == len(ref)
        zero = np.zeros(len(ref), np.int64)
        frac, idx = match_detections(mine.pseudo_boxes.tensor.cpu(), zero, ref.pseudo_boxes.tensor, zero, box_tol=5e-3)
        assert frac >= 0.97, f"pseudo boxes matched {frac:.3f}"
        ok = idx >= 0
        close(mine.scores_logists.cpu()[idx[ok]], ref.scores_logists[ok], *logit_tol, "pseudo logits")
        close(mine.boxes_sigma.cpu()[idx[ok]], ref.boxes_sigma[ok], *logit_tol, "pseudo sigma")


SUP = [k + "_sup" for k in ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg")]
UNSUP = [k + "_unsup" for k in ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg")]


def test_baseline_config2_full_mutual_learning_step_1333x800_vs_oracle(monkeypatch, capsys):
    """BASELINE.json configs[2] shape: final_c2f.yaml, BURN_UP_STEP = 0 -> EMA copy, teacher forward + pseudo labels,
    shrink-paste, joint supervised + unsupervised student pass, one backward, clip + SGD, with 1 labelled + 1 unlabelled
    1333 x 800 image.  The student of the oracle is handed the HIP teacher's pseudo labels (so that all eight student
    losses are compared on identical targets); the two teachers' pseudo labels are compared with each other."""
    res = mutual_learning_step_vs_oracle(monkeypatch, "configs/pt/final_c2f.yaml", 800, 1333)
    m, om = res["m"], res["om"]
    check_teacher_and_pseudo_labels(res)
    assert set(SUP + UNSUP) <= set(m) and set(SUP + UNSUP) <= set(om)
    with capsys.disabled():
        print(f"\n[configs[2] 1333x800] proposals: {res['log'].check_sets()}; pseudo labels {[len(p) for p in res['tr'].mine]}; "
              f"losses HIP {m} oracle {om}")
    _compare_step(m, om, res["tr"], res["state"], res["params"], SUP + UNSUP, "configs[2]")


def test_baseline_config2_step_b4_plus_4_1333x800_vs_oracle(monkeypatch, capsys):
    """configs[2] with 4 labelled + 4 unlabelled 1333 x 800 images: the joint 12-image student pass (8 labelled views + 4
    unlabelled strong views), a 4-image teacher pass, matcher / sampler / NMS batched over several images with different gt
    counts, per-image pseudo labels -- the batched code paths of the bench (16 + 16) against the oracle's per-image loops.
    Same bar as the 1 + 1 test: 8 losses 1e-4, gradient norm 1e-3, updated-parameter probes 1e-4."""
    res = mutual_learning_step_vs_oracle(monkeypatch, "configs/pt/final_c2f.yaml", 800, 1333, n_img=4, seed=47)
    m, om = res["m"], res["om"]
    check_teacher_and_pseudo_labels(res)
    assert set(SUP + UNSUP) <= set(m) and set(SUP + UNSUP) <= set(om)
    with capsys.disabled():
        print(f"\n[configs[2] 1333x800 B=4+4] proposals: {res['log'].check_sets()}; pseudo labels {[len(p) for p in res['tr'].mine]}; "
              f"losses HIP {m} oracle {om}")
    _compare_step(m, om, res["tr"], res["state"], res["params"], SUP + UNSUP, "configs[2] B=4+4")


def test_baseline_config3_per_gpu_share_b8_plus_8_1333x800_vs_oracle(monkeypatch, capsys):
    """configs[3]'s per-GPU share (batch 64 + 64 on 8 GPUs = 8 labelled + 8 unlabelled 1333 x 800 images per rank; VERDICT r4 item
    3c): the joint 24-image student pass (n = 24 launch shapes of every conv kernel, incl. the F(4x4,3x3) kernel of round 5), an
    8-image teacher pass, matcher / sampler / NMS batched over 8 / 16 / 24 images with different gt counts, 8 per-image pseudo-label
    sets -- against the oracle's per-image loops (reference step pt/engine/trainer.py:263-392, RPN terms
    pt/modeling/proposal_generator/rpn.py:257-361, ROI terms pt/modeling/roi_heads/fast_rcnn.py:179-263).
    Same bar as the 1 + 1 and 4 + 4 tests: 8 losses 1e-4, gradient norm 1e-3, updated-parameter probes 1e-4."""
    res = mutual_learning_step_vs_oracle(monkeypatch, "configs/pt/final_c2f.yaml", 800, 1333, n_img=8, seed=83)
    m, om = res["m"], res["om"]
    check_teacher_and_pseudo_labels(res)
    assert set(SUP + UNSUP) <= set(m) and set(SUP + UNSUP) <= set(om)
    with capsys.disabled():
        print(f"\n[configs[3] per-GPU share 1333x800 B=8+8] proposals: {res['log'].check_sets()}; pseudo labels "
              f"{[len(p) for p in res['tr'].mine]}; losses HIP {m} oracle {om}")
    _compare_step(m, om, res["tr"], res["state"], res["params"], SUP + UNSUP, "configs[3] per-GPU share B=8+8")


def test_baseline_config2_bench_batch_b16_plus_16_1333x800_vs_oracle(monkeypatch, capsys):
    """configs[2] at the BENCH's batch (16 labelled + 16 unlabelled 1333 x 800 images on one GPU; VERDICT r5 weak #3 / next-round
    item 7): the joint 48-image student pass, the 16-image teacher pass, matcher / sampler / NMS / pseudo-label relabelling batched
    over 16 / 32 / 48 images -- the exact launch shapes `bench.py` times -- against the oracle's per-image loops.  Same bar as the
    smaller steps: 8 losses 1e-4, gradient norm 1e-3, updated-parameter probes 1e-4; proposals exact up to the bounded tie
    permutation (MAX_TIED_ROWS_PERMUTED).  The oracle keeps the activations of a 48-image fp32 backward in host memory
    (~ 2.6 GB per image and pass): skipped, with the reason, on a host with < 400 GB available."""
    import psutil
    avail = psutil.virtual_memory().available / 2 ** 30
    if avail < 400:
        pytest.skip(f"the CPU oracle's 48-image backward needs ~ 300 GB of host memory; {avail:.0f} GB available "
                    "(the 8 + 8 step above covers the same code paths at n = 24)")
    res = mutual_learning_step_vs_oracle(monkeypatch, "configs/pt/final_c2f.yaml", 800, 1333, n_img=16, seed=161)
    m, om = res["m"], res["om"]
    check_teacher_and_pseudo_labels(res)
    assert set(SUP + UNSUP) <= set(m) and set(SUP + UNSUP) <= set(om)
    with capsys.disabled():
        print(f"\n[configs[2] bench batch 1333x800 B=16+16] proposals: {res['log'].check_sets()}; pseudo labels "
              f"{[len(p) for p in res['tr'].mine]}; losses HIP {m} oracle {om}")
    _compare_step(m, om, res["tr"], res["state"], res["params"], SUP + UNSUP, "configs[2] bench batch B=16+16")
