"""GPU parity tests (pytest -m gpu) of the fused Winograd F(4x4,3x3) kernel (csrc/wino4.hip, round 5) through the C ABI: every
epilogue, forward and dgrad packs, against stock torch CPU fp32 conv2d.

Tolerance: 1e-4 relative + 1e-4 absolute on unit-scale outputs -- the SAME bar as the F(2x2,3x3) and direct kernels' tests
(tests/test_wino_gpu.py); the kernel's interpolation points were chosen so that it passes it with margin (measured ~1e-5)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def g(seed):
    return torch.Generator().manual_seed(seed)


def close(a, b, rtol, atol, what=""):
    a = a.detach().cpu().double().numpy()
    b = b.detach().cpu().double().numpy()
    assert a.shape == b.shape, f"{what}: {a.shape} vs {b.shape}"
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), f"{what}: max abs err {err.max():.3e}, max |ref| {np.abs(b).max():.3e}, " \
                               f"{int((err > tol).sum())} of {err.size} off, first at {np.argwhere(err > tol)[0]}"
    return float(err.max())


FAMS = ["wino4", "wino4p"]        # csrc/wino4.hip (round 5) / csrc/wino4p.hip (round 6: positions split over the wave pair) -- same contract
FAM = "wino4"


@pytest.fixture(params=FAMS)
def fam(request):
    global FAM
    FAM = request.param
    yield request.param
    FAM = "wino4"


def _wino4(x, wt, bias, mask, epilogue, mode=0):
    """ptmi_conv3x3_<FAM>_pack_weights + ptmi_conv3x3_<FAM>_fwd on device tensors"""
    from probabilisticteacher_amd import _lib, ops
    call, ptr, stream = _lib.call, ops._ptr, ops._stream
    co, ci = wt.shape[0], wt.shape[1]
    conv_cin, conv_cout = (ci, co) if mode == 0 else (co, ci)
    n, cin, h, w = x.shape
    assert cin == conv_cin
    wp = torch.empty(getattr(_lib.load(), f"ptmi_conv3x3_{FAM}_packed_floats")(conv_cin, conv_cout), device=DEV)
    call(f"ptmi_conv3x3_{FAM}_pack_weights", ptr(wt), ptr(wp), co, ci, mode, stream())
    y = torch.full((n, conv_cout, h // 2, w // 2) if epilogue == 4 else (n, conv_cout, h, w), float("nan"), device=DEV)
    call(f"ptmi_conv3x3_{FAM}_fwd", ptr(x), ptr(wp), ptr(bias), ptr(mask), ptr(y), n, conv_cin, conv_cout, h, w, epilogue,
         stream())
    return y


SHAPES = [
    (1, 8, 64, 8, 64),        # two chunks, one workgroup, exact tile
    (1, 16, 64, 8, 64),       # four chunks (stage rotation)
    (1, 40, 64, 8, 64),       # ten chunks: every residue of the 3 slab stages / 4 patch stages
    (2, 64, 64, 24, 72),      # several tiles, second tile column 8 wide
    (2, 64, 128, 19, 35),     # odd H and W: tiles cut by the right edge, rows below the image
    (1, 128, 256, 13, 33),
    (1, 256, 512, 9, 83),     # W = 83 as the 1333x800 block-5 map
    (1, 24, 70, 11, 17),      # output channel count that is not a multiple of the channel tile
    (1, 32, 128, 5, 166),     # W = 166 / 333: 16-B pieces straddling the right image edge
    (2, 32, 128, 3, 333),     # H = 3: the lower wave pair of every workgroup is idle
    (1, 40, 130, 1, 70),      # a single row
    (3, 32, 64, 4, 3),        # narrower than one 16-B piece
    (1, 8, 7, 2, 1),          # one column
    (1, 24, 64, 50, 83),      # the block-5 map itself
    (3, 16, 48, 7, 21),       # workgroups straddling bands and images
]


@pytest.mark.parametrize("n,cin,cout,h,w", SHAPES)
def test_wino4_forward_epilogues(fam, n, cin, cout, h, w):
    gen = g(n * 1000 + cin + cout + h + w)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    ref = F.conv2d(x, wt, b, padding=1)
    xd, wd, bd = x.to(DEV), wt.to(DEV), b.to(DEV)
    close(_wino4(xd, wd, bd, None, 0), ref, 1e-4, 1e-4, "epilogue 0 (bias)")
    close(_wino4(xd, wd, bd, None, 1), F.relu(ref), 1e-4, 1e-4, "epilogue 1 (bias + relu)")
    close(_wino4(xd, wd, None, None, 2), ref - b.view(1, -1, 1, 1), 1e-4, 1e-4, "epilogue 2 (none)")
    if h >= 2 and w >= 2:
        close(_wino4(xd, wd, bd, None, 4), F.max_pool2d(F.relu(ref), 2, 2), 1e-4, 1e-4, "epilogue 4 (bias + relu + pool)")


@pytest.mark.parametrize("n,cin,cout,h,w", SHAPES)
def test_wino4_dgrad_with_relu_mask(fam, n, cin, cout, h, w):
    """dX = conv(dY, W^T flipped) (pack mode 1), plain (epilogue 2) and through the producer's ReLU mask (epilogue 3).
    (The conv's input channel count is `cout` here: only shapes whose cout is a multiple of 8 are served.)"""
    if cout % 8:
        pytest.skip("dgrad contracts over cout: not a multiple of 8")
    gen = g(7 + n * 1000 + cin + cout + h + w)
    xr = torch.randn(n, cin, h, w, generator=gen).requires_grad_()
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    gy = torch.randn(n, cout, h, w, generator=gen)
    F.conv2d(xr, wt, None, padding=1).backward(gy)
    mask_src = torch.randn(n, cin, h, w, generator=gen)            # "activation of the producing layer"
    got = _wino4(gy.to(DEV), wt.to(DEV), None, None, 2, mode=1)
    close(got, xr.grad, 1e-4, 2e-4, "dgrad")
    got3 = _wino4(gy.to(DEV), wt.to(DEV), None, mask_src.to(DEV), 3, mode=1)
    close(got3, xr.grad * (mask_src > 0), 1e-4, 2e-4, "dgrad + relu mask")


def test_wino4_rejects_channel_counts_it_does_not_serve(fam):
    from probabilisticteacher_amd import _lib
    lib = _lib.load()
    fits = getattr(lib, f"ptmi_conv3x3_{fam}_fwd_fits")
    assert fits(20, 64, 8, 8) == 0 and fits(24, 64, 8, 8) == 1
    x = torch.zeros(1, 20, 8, 8, device=DEV)
    with pytest.raises(_lib.PtmiError):
        _wino4(x, torch.zeros(64, 20, 3, 3, device=DEV), None, None, 2)


def test_wino4_baseline_layer_shapes(fam):
    """One image of every distinct layer shape of the 1333x800 stack that the F(4x4,3x3) kernel may serve, against torch CPU on
    border-including crops, + the measured error (printed: the margin to the 1e-4 bar)."""
    worst = 0.0
    for cin, cout, h, w in [(64, 64, 800, 1333), (128, 128, 400, 666), (256, 256, 200, 333), (512, 512, 100, 166),
                            (512, 512, 50, 83)]:
        gen = g(cin + h)
        x = torch.relu(torch.randn(1, cin, h, w, generator=gen))
        wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
        b = torch.randn(cout, generator=gen) * 0.1
        got = _wino4(x.to(DEV), wt.to(DEV), b.to(DEV), None, 1).cpu()
        assert torch.isfinite(got).all()
        for ys, xs in [(slice(0, 24), slice(0, 40)), (slice(h - 24, h), slice(w - 40, w)), (slice(h // 2 - 9, h // 2 + 9), slice(w - 45, w))]:
            y0, y1 = max(ys.start - 1, 0), min(ys.stop + 1, h)
            x0, x1 = max(xs.start - 1, 0), min(xs.stop + 1, w)
            ref = F.relu(F.conv2d(x[:, :, y0:y1, x0:x1], wt, b, padding=1))
            ref = ref[:, :, ys.start - y0: ys.start - y0 + (ys.stop - ys.start), xs.start - x0: xs.start - x0 + (xs.stop - xs.start)]
            worst = max(worst, close(got[:, :, ys, xs], ref, 1e-4, 1e-4, f"layer {cin}->{cout} {h}x{w} crop {ys} {xs}"))
    print(f"\n[{fam}] worst abs error over the layer-shape crops: {worst:.2e} (bar 1e-4 + 1e-4 |ref|)")


# ------------------------------------------------------------------------------------------------ weight gradient (csrc/wino4w.hip)
def _wino4_wgrad(x, gy, cout, accumulate_into=None):
    from probabilisticteacher_amd import _lib, ops
    n, cin, h, w = x.shape
    dw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV) if accumulate_into is None else accumulate_into[0]
    db = torch.full((cout,), float("nan"), device=DEV) if accumulate_into is None else accumulate_into[1]
    ws = torch.empty(_lib.load().ptmi_conv3x3_wino4_wgrad_ws_floats(n, cin, cout, h, w), device=DEV)
    _lib.call("ptmi_conv3x3_wino4_wgrad", ops._ptr(x), ops._ptr(gy), ops._ptr(dw), ops._ptr(db), ops._ptr(ws), n, cin, cout, h, w,
              0 if accumulate_into is None else 1, ops._stream())
    return dw, db


WG_SHAPES = [
    (1, 32, 64, 4, 16),       # one chunk, one workgroup
    (1, 32, 64, 8, 32),       # four chunks: stage rotation
    (2, 64, 64, 24, 72),      # several chunks, two ci tiles
    (2, 64, 128, 19, 35),     # odd H and W: border chunks (rows below the image, pieces straddling the right edge)
    (1, 128, 256, 13, 33),
    (1, 256, 512, 9, 83),     # W = 83 as the 1333x800 block-5 map
    (1, 40, 70, 11, 17),      # channel counts that are not multiples of the tiles
    (1, 32, 128, 5, 166),
    (2, 32, 128, 3, 333),     # H = 3: every chunk is a border chunk
    (1, 40, 130, 1, 70),      # a single row
    (3, 32, 64, 4, 3),        # narrower than one 16-B piece
    (1, 24, 64, 50, 83),      # the block-5 map itself
    (3, 16, 48, 7, 21),
    (2, 64, 64, 40, 100),     # interior (plain) chunks on all sides
]


@pytest.mark.parametrize("n,cin,cout,h,w", WG_SHAPES)
def test_wino4_wgrad(n, cin, cout, h, w):
    """dW, db of ptmi_conv3x3_wino4_wgrad against torch CPU fp32 autograd: 1e-4 relative + 1e-4 of the gradient's scale (the bar of
    the F(2x2,3x3)-domain and direct kernels' tests); bit-identical repeats; accumulate = 1 adds"""
    gen = g(31 + n * 1000 + cin + cout + h + w)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = (torch.randn(cout, cin, 3, 3, generator=gen) * 0.1).requires_grad_()
    b = torch.zeros(cout, requires_grad=True)
    gy = torch.randn(n, cout, h, w, generator=gen)
    F.conv2d(x, wt, b, padding=1).backward(gy)
    dw, db = _wino4_wgrad(x.to(DEV), gy.to(DEV), cout)
    scale = float(wt.grad.abs().max())
    close(dw, wt.grad, 1e-4, 1e-4 * scale, "dW")
    close(db, b.grad, 1e-4, 1e-4 * float(b.grad.abs().max()), "db")
    dw2, db2 = _wino4_wgrad(x.to(DEV), gy.to(DEV), cout)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "the split reduction runs in a fixed order"
    acc = (dw.clone(), db.clone())
    _wino4_wgrad(x.to(DEV), gy.to(DEV), cout, accumulate_into=acc)
    close(acc[0], 2 * wt.grad, 1e-4, 2e-4 * scale, "dW accumulate")
    close(acc[1], 2 * b.grad, 1e-4, 2e-4 * float(b.grad.abs().max()), "db accumulate")


def test_wino4_wgrad_layer_shape_vs_direct_kernel():
    """A trainable layer at the 1333x800 map size (conv4: 256 -> 512 at 100x166, 2 images) against the direct split-K kernel"""
    from probabilisticteacher_amd import _lib, ops
    gen = g(5)
    x = torch.randn(2, 256, 100, 166, generator=gen).to(DEV)
    gy = torch.randn(2, 512, 100, 166, generator=gen).to(DEV)
    dw, db = _wino4_wgrad(x, gy, 512)
    ws = torch.empty(_lib.load().ptmi_conv3x3_wgrad_ws_floats(2, 256, 512, 100, 166), device=DEV)
    dw_d, db_d = torch.empty_like(dw), torch.empty_like(db)
    _lib.call("ptmi_conv3x3_wgrad", ops._ptr(x), ops._ptr(gy), ops._ptr(dw_d), ops._ptr(db_d), ops._ptr(ws), 2, 256, 512,
              100, 166, 0, ops._stream())
    scale = float(dw_d.abs().max())
    close(dw, dw_d, 1e-4, 1e-4 * scale, "dW wino4 vs direct")
    close(db, db_d, 1e-4, 1e-4 * float(db_d.abs().max()), "db")


# ---- round 6: the dynamic tile schedule (ptmi_conv3x3_wino4_fwd_sched) ------------------------------------------------------
def _wino4_sched(x, wp, bias, mask, epilogue, conv_cout, sched):
    from probabilisticteacher_amd import _lib, ops
    n, cin, h, w = x.shape
    y = torch.full((n, conv_cout, h // 2, w // 2) if epilogue == 4 else (n, conv_cout, h, w), float("nan"), device=DEV)
    _lib.call(f"ptmi_conv3x3_{FAM}_fwd_sched", ops._ptr(x), ops._ptr(wp), ops._ptr(bias), ops._ptr(mask), ops._ptr(y), n, cin,
              conv_cout, h, w, epilogue, ops._ptr(sched), ops._stream())
    return y


SCHED_SHAPES = SHAPES + [
    (6, 64, 64, 100, 166),     # one channel tile, 1014 pixel tiles: ~4 tiles per workgroup, nPix % 8 != 0 (colocate tail ids invalid)
    (4, 64, 256, 50, 83),      # four channel tiles (colocated on one queue each), 155 pixel tiles
    (3, 128, 512, 50, 83),     # eight channel tiles: queue q = channel tile q
    (1, 64, 192, 8, 64),       # fewer tiles (3) than queues
    (2, 64, 640, 30, 70),      # ten channel tiles
]


@pytest.mark.parametrize("n,cin,cout,h,w", SCHED_SHAPES)
def test_wino4_dynamic_schedule_is_bit_identical_to_static_and_rearms(fam, n, cin, cout, h, w):
    """The work-queue schedule hands every tile to exactly one workgroup: the output equals the static walk's BIT FOR BIT (a tile's
    arithmetic does not depend on who computes it; a tile drawn twice would also pass, a tile never drawn leaves the NaN fill), for
    every epilogue, and the 16 schedule words are zero again after each launch -- the same buffer serves the next launch."""
    from probabilisticteacher_amd import _lib, ops
    gen = g(n * 977 + cin + cout + h + w)
    x = torch.randn(n, cin, h, w, generator=gen).to(DEV)
    wt = (torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))).to(DEV)
    b = (torch.randn(cout, generator=gen) * 0.1).to(DEV)
    wp = torch.empty(getattr(_lib.load(), f"ptmi_conv3x3_{FAM}_packed_floats")(cin, cout), device=DEV)
    _lib.call(f"ptmi_conv3x3_{FAM}_pack_weights", ops._ptr(wt), ops._ptr(wp), cout, cin, 0, ops._stream())
    mask = torch.randn(n, cout, h, w, generator=gen).to(DEV)
    sched = torch.zeros(16, dtype=torch.int32, device=DEV)
    for epi, bias, m in ((1, b, None), (2, None, None), (3, None, mask), (4, b, None)):
        if epi == 4 and (h < 2 or w < 2):
            continue
        ref = _wino4_sched(x, wp, bias, m, epi, cout, None)                 # static
        assert bool(torch.isfinite(ref).all())
        for rep in range(3):                                                # the buffer re-arms itself
            got = _wino4_sched(x, wp, bias, m, epi, cout, sched)
            assert torch.equal(got, ref), f"epilogue {epi} launch {rep}: dynamic schedule differs from the static walk"
            assert not bool(sched.any()), f"epilogue {epi} launch {rep}: schedule words not zero after the launch: {sched.tolist()}"


def test_wino4_dynamic_schedule_under_cu_contention(fam):
    """A side stream holds 64 CUs' worth of LDS while the convolution launches (what an RCCL kernel overlapping backward does):
    the output stays bit-identical and the schedule words re-arm.  (Timing under contention: tools/exp/contention.py.)"""
    from probabilisticteacher_amd import _lib, ops
    gen = g(77)
    n, cin, cout, h, w = 8, 64, 128, 100, 166
    x = torch.randn(n, cin, h, w, generator=gen).to(DEV)
    wt = (torch.randn(cout, cin, 3, 3, generator=gen) * 0.04).to(DEV)
    b = (torch.randn(cout, generator=gen) * 0.1).to(DEV)
    wp = torch.empty(getattr(_lib.load(), f"ptmi_conv3x3_{FAM}_packed_floats")(cin, cout), device=DEV)
    _lib.call(f"ptmi_conv3x3_{FAM}_pack_weights", ops._ptr(wt), ops._ptr(wp), cout, cin, 0, ops._stream())
    sched = torch.zeros(16, dtype=torch.int32, device=DEV)
    ref = _wino4_sched(x, wp, b, None, 1, cout, None)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    for rep in range(3):
        with torch.cuda.stream(side):
            _lib.call("ptmi_hold_cus", ops._ptr(flag), 64, 20000, ctypes_stream(side))
        got = _wino4_sched(x, wp, b, None, 1, cout, sched)
        torch.cuda.synchronize()
        assert torch.equal(got, ref) and not bool(sched.any())


def ctypes_stream(s):
    import ctypes
    return ctypes.c_void_p(s.cuda_stream)


@pytest.mark.parametrize("name", ["ptmi_conv3x3_wino4_wgrad", "ptmi_conv3x3_wino_wgrad"])
@pytest.mark.parametrize("waves", [1, 3, 4, 16])      # 3: PTrainer.ddp_wgrad_waves(False)
def test_wgrad_waves_entry_points(name, waves):
    """round 6: the Winograd-domain weight-gradient kernels with `waves` fills of the chip (what PTrainer selects under DDP): same
    result as one fill up to the summation order of the split partials (1e-4 of the scale vs float64 torch; bitwise repeatable)"""
    from probabilisticteacher_amd import _lib, ops
    lib = _lib.load()
    gen = g(300 + waves)
    n, cin, cout, h, w = 3, 64, 128, 37, 70
    x = torch.relu(torch.randn(n, cin, h, w, generator=gen))
    gy = torch.randn(n, cout, h, w, generator=gen)
    wr = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    br = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wr, br, padding=1).backward(gy.double())
    xd, gd = x.to(DEV), gy.to(DEV)
    outs = []
    for _ in range(2):
        dw, db = torch.empty(cout, cin, 3, 3, device=DEV), torch.empty(cout, device=DEV)
        ws = torch.empty(getattr(lib, name + "_ws_floats_waves")(n, cin, cout, h, w, waves), device=DEV)
        _lib.call(name + "_waves", ops._ptr(xd), ops._ptr(gd), ops._ptr(dw), ops._ptr(db), ops._ptr(ws), n, cin, cout, h, w, 0, waves,
                  ops._stream())
        outs.append((dw.clone(), db.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "fixed-order reduction of the split partials"
    assert float((outs[0][0].cpu().double() - wr.grad).abs().max()) <= 1e-4 * float(wr.grad.abs().max())
    assert float((outs[0][1].cpu().double() - br.grad).abs().max()) <= 1e-4 * float(br.grad.abs().max())
    assert getattr(lib, name + "_ws_floats_waves")(n, cin, cout, h, w, 1) == getattr(lib, name + "_ws_floats")(n, cin, cout, h, w)
