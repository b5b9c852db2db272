"""GPU parity tests (pytest -m gpu) of the fused Winograd F(2x2,3x3) kernel (csrc/wino.hip) through the C ABI:
every epilogue, forward and dgrad packs, against stock torch CPU fp32 conv2d AND against the direct kernel of conv.hip.

Tolerance: 1e-4 relative + 1e-4 of the tensor's scale -- the same bar as the direct kernel's tests (the transforms add a
few fp32 roundings per output; measured differences are ~1e-6 of the scale)."""
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


def _wino(x, wt, bias, mask, epilogue, mode=0):
    """ptmi_conv3x3_wino_pack_weights + ptmi_conv3x3_wino_fwd on device tensors"""
    from probabilisticteacher_amd import _lib, ops
    call, ptr, stream = _lib.call, ops._ptr, ops._stream
    co, ci = wt.shape[0], wt.shape[1]
    conv_cin, conv_cout = (ci, co) if mode == 0 else (co, ci)
    n, cin, h, w = x.shape
    assert cin == conv_cin
    wp = torch.empty(_lib.load().ptmi_conv3x3_wino_packed_floats(conv_cin, conv_cout), device=DEV)
    call("ptmi_conv3x3_wino_pack_weights", ptr(wt), ptr(wp), co, ci, mode, stream())
    y = torch.full((n, conv_cout, h // 2, w // 2) if epilogue == 4 else (n, conv_cout, h, w), float("nan"), device=DEV)
    call("ptmi_conv3x3_wino_fwd", ptr(x), ptr(wp), ptr(bias), ptr(mask), ptr(y), n, conv_cin, conv_cout, h, w, epilogue,
         stream())
    return y


SHAPES = [
    (1, 8, 64, 8, 32),        # one chunk, one workgroup, exact tile
    (1, 16, 64, 8, 32),       # two chunks (double buffer hand-over)
    (2, 64, 64, 24, 40),      # several tiles, second tile column 8 wide
    (2, 64, 128, 19, 35),     # odd H and W: single-column lane at the right edge, rows below the image
    (1, 128, 256, 13, 33),
    (1, 256, 512, 9, 83),     # W = 83 as the 1333x800 block-5 map
    (1, 20, 70, 11, 17),      # channel counts that are not multiples of the chunk / channel tile
    (1, 32, 128, 5, 166),     # W = 166 / 333: 16-B pieces straddling the right image edge
    (2, 32, 128, 3, 333),     # H = 3: the lower wave pair of every workgroup is idle
    (1, 40, 130, 1, 70),      # a single row
    (3, 32, 64, 4, 3),        # narrower than one 16-B piece
    (1, 5, 7, 2, 1),          # one column
    (1, 24, 64, 50, 83),      # the block-5 map itself (7 x 3 workgroups, ragged both ways)
]


@pytest.mark.parametrize("n,cin,cout,h,w", SHAPES)
def test_wino_forward_epilogues(n, cin, cout, h, w):
    gen = g(n * 1000 + cin + cout + h + w)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    ref = F.conv2d(x, wt, b, padding=1)
    xd, wd, bd = x.to(DEV), wt.to(DEV), b.to(DEV)
    close(_wino(xd, wd, bd, None, 0), ref, 1e-4, 1e-4, "epilogue 0 (bias)")
    close(_wino(xd, wd, bd, None, 1), F.relu(ref), 1e-4, 1e-4, "epilogue 1 (bias + relu)")
    close(_wino(xd, wd, None, None, 2), ref - b.view(1, -1, 1, 1), 1e-4, 1e-4, "epilogue 2 (none)")
    if h >= 2 and w >= 2:
        close(_wino(xd, wd, bd, None, 4), F.max_pool2d(F.relu(ref), 2, 2), 1e-4, 1e-4, "epilogue 4 (bias + relu + pool)")


@pytest.mark.parametrize("n,cin,cout,h,w", SHAPES)
def test_wino_dgrad_with_relu_mask(n, cin, cout, h, w):
    """dX = conv(dY, W^T flipped) (pack mode 1), plain (epilogue 2) and through the producer's ReLU mask (epilogue 3)."""
    gen = g(7 + n * 1000 + cin + cout + h + w)
    xr = torch.randn(n, cin, h, w, generator=gen).requires_grad_()
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    gy = torch.randn(n, cout, h, w, generator=gen)
    F.conv2d(xr, wt, None, padding=1).backward(gy)
    mask_src = torch.randn(n, cin, h, w, generator=gen)            # "activation of the producing layer"
    got = _wino(gy.to(DEV), wt.to(DEV), None, None, 2, mode=1)
    close(got, xr.grad, 1e-4, 2e-4, "dgrad")
    got3 = _wino(gy.to(DEV), wt.to(DEV), None, mask_src.to(DEV), 3, mode=1)
    close(got3, xr.grad * (mask_src > 0), 1e-4, 2e-4, "dgrad + relu mask")


def test_wino_ragged_channels_do_not_read_the_next_image():
    """Cin not a multiple of the 8-channel chunk: the last chunk's padding channels must be zeros, not the next image's first
    planes (whose packed weights are zero -- but 0 x NaN is NaN).  Image 1 is all NaN; image 0's output must be unaffected."""
    gen = g(99)
    x = torch.randn(2, 20, 9, 35, generator=gen)
    wt = torch.randn(70, 20, 3, 3, generator=gen) * 0.1
    b = torch.randn(70, generator=gen) * 0.1
    ref = F.relu(F.conv2d(x[:1], wt, b, padding=1))
    x[1] = float("nan")
    got = _wino(x.to(DEV), wt.to(DEV), b.to(DEV), None, 1)
    close(got[:1], ref, 1e-4, 1e-4, "image 0 next to a NaN image")
    # (image 1 itself: the fused ReLU is v_max_f32, which returns 0 for NaN where torch.relu propagates it -- non-finite
    # activations are outside the parity domain; the heads' outputs pass through no ReLU and the step aborts on them as the
    # reference does)
    got2 = _wino(x.to(DEV), wt.to(DEV), b.to(DEV), None, 0)
    assert torch.isnan(got2[1]).all() and torch.isfinite(got2[0]).all()


def test_wino_equals_direct_kernel_closely():
    """Same inputs through the algorithms (ops.set_conv_algo): forward and dgrad of F(2x2,3x3) ("wino2") agree with the direct
    kernel to fp32 rounding (1e-5); F(4x4,3x3) ("auto" since round 5 for this channel count) agrees at the parity bar of every
    conv test, 1e-4 (its transforms multiply by constants up to 3.4: measured 1-3e-5, tests/test_wino4_gpu.py)."""
    from probabilisticteacher_amd import ops
    gen = g(3)
    x = torch.randn(2, 64, 45, 70, generator=gen).to(DEV)
    wt = (torch.randn(128, 64, 3, 3, generator=gen) * 0.06).to(DEV)
    b = (torch.randn(128, generator=gen) * 0.1).to(DEV)
    gy = torch.randn(2, 128, 45, 70, generator=gen).to(DEV)
    outs = {}
    try:
        for algo in ("auto", "wino2", "direct"):
            ops.set_conv_algo(algo)
            xd = x.clone().requires_grad_()
            y = ops.conv3x3(xd, wt, b, True)
            y.backward(gy)
            outs[algo] = (y.detach(), xd.grad)
    finally:
        ops.set_conv_algo("auto")
    close(outs["wino2"][0], outs["direct"][0], 1e-5, 2e-5, "forward wino vs direct")
    close(outs["wino2"][1], outs["direct"][1], 1e-5, 2e-5, "dgrad wino vs direct")
    close(outs["auto"][0], outs["direct"][0], 1e-4, 1e-4, "forward wino4 vs direct")
    close(outs["auto"][1], outs["direct"][1], 1e-4, 1e-4, "dgrad wino4 vs direct")


def test_wino_baseline_layer_shapes():
    """One image of every distinct Winograd layer shape of the 1333x800 stack against torch CPU on border-including crops
    (the full-size comparison of every layer runs in test_baseline_size_gpu.py through the production autograd nodes)."""
    for cin, cout, h, w in [(64, 64, 800, 1333), (128, 256, 200, 333), (512, 512, 50, 83)]:
        gen = g(cin + h)
        x = torch.randn(1, cin, h, w, generator=gen)
        wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
        b = torch.randn(cout, generator=gen) * 0.1
        got = _wino(x.to(DEV), wt.to(DEV), b.to(DEV), None, 1).cpu()
        assert torch.isfinite(got).all()
        for ys, xs in [(slice(0, 24), slice(0, 40)), (slice(h - 24, h), slice(w - 40, w)), (slice(h // 2 - 9, h // 2 + 9), slice(w - 45, w))]:
            # a crop grown by one pixel of context on each interior side gives exact values for the crop's interior
            y0, y1 = max(ys.start - 1, 0), min(ys.stop + 1, h)
            x0, x1 = max(xs.start - 1, 0), min(xs.stop + 1, w)
            ref = F.relu(F.conv2d(x[:, :, y0:y1, x0:x1], wt, b, padding=1))
            ref = ref[:, :, ys.start - y0: ys.start - y0 + (ys.stop - ys.start), xs.start - x0: xs.start - x0 + (xs.stop - xs.start)]
            close(got[:, :, ys, xs], ref, 1e-4, 1e-4, f"layer {cin}->{cout} {h}x{w} crop {ys} {xs}")


def _wino_wgrad(x, gy, cout, accumulate_into=None):
    from probabilisticteacher_amd import _lib, ops
    call, ptr, stream = _lib.call, ops._ptr, ops._stream
    n, cin, h, w = x.shape
    ws = torch.full((_lib.load().ptmi_conv3x3_wino_wgrad_ws_floats(n, cin, cout, h, w),), float("nan"), device=DEV)
    dw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV) if accumulate_into is None else accumulate_into[0]
    db = torch.full((cout,), float("nan"), device=DEV) if accumulate_into is None else accumulate_into[1]
    call("ptmi_conv3x3_wino_wgrad", ptr(x), ptr(gy), ptr(dw), ptr(db), ptr(ws), n, cin, cout, h, w,
         0 if accumulate_into is None else 1, stream())
    return dw, db


WGRAD_SHAPES = SHAPES + [
    (2, 64, 64, 6, 64),       # exact chunks: two column blocks, three tile rows
    (3, 128, 64, 7, 100),     # several chunks per split, odd H, ragged last column block
    (1, 64, 192, 2, 36),      # the last column block holds only the 4 rightmost columns
    (2, 70, 100, 9, 37),      # ragged channel tiles both sides
    (4, 64, 64, 40, 45),      # many chunks per split: the steady-state pipeline over both LDS stages
    (4, 64, 64, 40, 64),      # the same with 8 k-steps per chunk (W = 32, 64, 666, 1333 take KSN = 8, the others 7)
    (1, 64, 64, 6, 84),       # KSN = 7, three exact 28-column blocks, 9 chunks over 9 splits
    (5, 64, 64, 22, 56),      # KSN = 7, odd chunk counts per split (one empty chunk closes the two-chunk loop body)
]


@pytest.mark.parametrize("n,cin,cout,h,w", WGRAD_SHAPES)
def test_wino_wgrad(n, cin, cout, h, w):
    """dW, db of the Winograd-domain weight gradient against torch CPU fp32 autograd.  Tolerance: 1e-4 of the gradient's
    scale + 1e-4 relative (K = N H W products per weight; the transform-domain sums differ from direct summation by fp32
    rounding only)."""
    gen = g(11 + n * 1000 + cin + cout + h + w)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = (torch.randn(cout, cin, 3, 3, generator=gen) * 0.05).requires_grad_()
    b = torch.zeros(cout, requires_grad=True)
    gy = torch.randn(n, cout, h, w, generator=gen)
    F.conv2d(x, wt, b, padding=1).backward(gy)
    dw, db = _wino_wgrad(x.to(DEV), gy.to(DEV), cout)
    scale = float(wt.grad.abs().max())
    close(dw, wt.grad, 1e-4, 1e-4 * scale, "dW")
    close(db, b.grad, 1e-4, 1e-4 * float(b.grad.abs().max()), "db")
    # accumulate = 1 adds to the existing gradient; identical launches are bit-identical (fixed reduction order)
    dw2, db2 = _wino_wgrad(x.to(DEV), gy.to(DEV), cout)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    acc = (dw.clone(), db.clone())
    _wino_wgrad(x.to(DEV), gy.to(DEV), cout, accumulate_into=acc)
    close(acc[0], 2 * wt.grad, 1e-4, 2e-4 * scale, "dW accumulate")


def test_wino_wgrad_layer_shape_vs_direct_kernel():
    """A trainable layer at the 1333x800 map size (conv4: 256 -> 512 at 100x166, 2 images) against the direct split-K kernel"""
    from probabilisticteacher_amd import _lib, ops
    gen = g(5)
    x = torch.randn(2, 256, 100, 166, generator=gen).to(DEV)
    gy = torch.randn(2, 512, 100, 166, generator=gen).to(DEV)
    dw, db = _wino_wgrad(x, gy, 512)
    ws = torch.empty(_lib.load().ptmi_conv3x3_wgrad_ws_floats(2, 256, 512, 100, 166), device=DEV)
    dw_d, db_d = torch.empty_like(dw), torch.empty_like(db)
    _lib.call("ptmi_conv3x3_wgrad", ops._ptr(x), ops._ptr(gy), ops._ptr(dw_d), ops._ptr(db_d), ops._ptr(ws), 2, 256, 512,
              100, 166, 0, ops._stream())
    scale = float(dw_d.abs().max())
    close(dw, dw_d, 1e-4, 1e-4 * scale, "dW wino vs direct")
    close(db, db_d, 1e-4, 1e-4 * float(db_d.abs().max()), "db")


def test_training_step_runs_on_the_winograd_kernels():
    """Routing guard: one fp32 VGG block forward + backward through the production autograd nodes launches the Winograd forward /
    dgrad and weight-gradient kernels (and no direct conv kernel), and their issued-FLOP models are the ones the bench reports.
    Round 5: "auto" routes forward / dgrad of these channel counts to the F(4x4,3x3) kernel, "wino2" to F(2x2,3x3)."""
    from probabilisticteacher_amd import ops
    for algo, key, issued in (("auto", "conv3x3_wino4", ops.wino4_issued_flops), ("wino2", "conv3x3_wino", ops.wino_issued_flops)):
        ops.set_conv_algo(algo)
        try:
            _routing_guard(ops, key, issued)
        finally:
            ops.set_conv_algo("auto")


def _routing_guard(ops, key, issued_flops):
    gen = g(21)
    x = torch.randn(2, 64, 40, 83, generator=gen).to(DEV).requires_grad_()
    w1 = (torch.randn(128, 64, 3, 3, generator=gen) * 0.05).to(DEV).requires_grad_()
    b1 = torch.zeros(128, device=DEV, requires_grad=True)
    w2 = (torch.randn(128, 128, 3, 3, generator=gen) * 0.05).to(DEV).requires_grad_()
    b2 = torch.zeros(128, device=DEV, requires_grad=True)
    ops.profile_start()
    y = ops.conv3x3(ops.conv3x3(x, w1, b1, True), w2, b2, True)
    y.sum().backward()
    prof = ops.profile_stop()
    assert prof[key]["calls"] == 4 and prof["conv3x3_wino_wgrad"]["calls"] == 2, prof.keys()
    assert "conv3x3_mfma" not in prof and "conv3x3_wgrad" not in prof
    assert prof[key]["issued"] == 2 * issued_flops(2, 64, 128, 40, 83) + 2 * issued_flops(2, 128, 128, 40, 83)
    assert prof["conv3x3_wino_wgrad"]["issued"] == ops.wino_wgrad_issued_flops(2, 64, 128, 40, 83) + ops.wino_wgrad_issued_flops(2, 128, 128, 40, 83)
    ref = F.relu(F.conv2d(F.relu(F.conv2d(x.detach().cpu(), w1.detach().cpu(), None, padding=1)), w2.detach().cpu(), None, padding=1))
    close(y, ref, 1e-4, 1e-4, "two-layer block")


def test_routing_guard_wino4_wgrad_from_the_autograd_node():
    """a map with chunk fill >= 0.9 (40 x 96: whole 4 x 16 chunks) sends the weight gradients of the autograd nodes to
    conv3x3_wino4_wgrad -- including the 32-input-channel layer (cin 32 .. 63 with cout >= 64 ran the direct kernel before round 5) --
    and the issued-FLOP model matches what the profile hooks record; values vs torch CPU at the 1e-4 bar."""
    from probabilisticteacher_amd import ops
    gen = g(22)
    h, w = 40, 96
    assert ops._wino4_wgrad_fill(h, w) >= ops._WINO4_WGRAD_MIN_FILL
    assert ops._wgrad_kind(32, 64, h, w) == "wino4w" and ops._wgrad_kind(64, 128, h, w) == "wino4w"
    assert ops._wgrad_kind(512, 512, 50, 83) == "wino", "round 5/6 routing of the 50 x 83 maps (fill 0.83)"
    x = torch.randn(2, 32, h, w, generator=gen).to(DEV).requires_grad_()
    w1 = (torch.randn(64, 32, 3, 3, generator=gen) * 0.07).to(DEV).requires_grad_()
    b1 = (torch.randn(64, generator=gen) * 0.1).to(DEV).requires_grad_()
    w2 = (torch.randn(128, 64, 3, 3, generator=gen) * 0.05).to(DEV).requires_grad_()
    b2 = (torch.randn(128, generator=gen) * 0.1).to(DEV).requires_grad_()
    gy = torch.randn(2, 128, h, w, generator=gen)
    ops.profile_start()
    a1 = ops.conv3x3(x, w1, b1, True)
    y = ops.conv3x3(a1, w2, b2, True)
    y.backward(gy.to(DEV))
    prof = ops.profile_stop()
    assert prof["conv3x3_wino4_wgrad"]["calls"] == 2 and "conv3x3_wino_wgrad" not in prof and "conv3x3_wgrad" not in prof, prof.keys()
    assert prof["conv3x3_wino4_wgrad"]["issued"] == (ops.wino4_wgrad_issued_flops(2, 32, 64, h, w) +
                                                     ops.wino4_wgrad_issued_flops(2, 64, 128, h, w))
    xr, w1r, b1r, w2r, b2r = (t.detach().cpu().requires_grad_() for t in (x, w1, b1, w2, b2))
    # the reference backward uses the HIP side's ReLU decisions (a pre-activation within rounding of 0 flips its mask bit between any
    # two fp32 implementations, and ONE flipped pixel moves a 7 680-pixel weight-gradient sum by ~1e-3 of its scale: first run of this test)
    m1, m2 = (a1.detach().cpu() > 0).float(), (y.detach().cpu() > 0).float()
    yr = F.conv2d(F.conv2d(xr, w1r, b1r, padding=1) * m1, w2r, b2r, padding=1) * m2
    yr.backward(gy)
    close(y, yr.detach(), 1e-4, 1e-4, "two-layer block")
    for name, a, b in (("dW1", w1.grad, w1r.grad), ("db1", b1.grad, b1r.grad), ("dW2", w2.grad, w2r.grad), ("db2", b2.grad, b2r.grad)):
        sc = float(b.abs().max())
        err = float((a.cpu() - b).abs().max())
        assert err <= 1e-4 * sc, f"{name}: max abs err {err:.3e} vs scale {sc:.3e}"


def test_map_beyond_the_winograd_32bit_offsets_runs_the_direct_kernel():
    """ptmi_conv3x3_wino_fwd / _wgrad reject maps whose per-workgroup byte offsets exceed 32 bits (about 8 M pixels at
    64 channels); ops routes such a layer to the direct kernels instead of surfacing the exception (ptmi_conv3x3_wino_fwd_fits /
    _wgrad_fits).  The window is narrow -- the direct kernel addresses one image through 32-bit offsets as well -- and widest for
    many channels: 128 -> 128 channels at 2000 x 2050 (4.1 M pixels: (2 x 128 + 8) channel planes exceed 4 GB, 128 + 128 do not):
    forward, dgrad (direct kernel) and weight gradient (the Winograd-domain kernel, which reaches 8 M pixels) vs torch CPU fp32."""
    from probabilisticteacher_amd import _lib, ops
    torch.set_num_threads(max(2, min(__import__("os").cpu_count() or 2, 64)))
    n, c, h, w = 1, 128, 2000, 2050
    lib = _lib.load()
    assert lib.ptmi_conv3x3_wino_fwd_fits(c, c, 1024, 2048) == 1 and lib.ptmi_conv3x3_wino_fwd_fits(c, c, h, w) == 0
    assert lib.ptmi_conv3x3_wino_wgrad_fits(h, w) == 1 and lib.ptmi_conv3x3_wino_wgrad_fits(4096, 4096) == 0
    assert ops._use_wino(c, c, (1024, 2048)) and not ops._use_wino(c, c, (h, w))
    x = torch.randn(n, c, h, w, generator=g(5))
    wt = torch.randn(c, c, 3, 3, generator=g(6)) * math.sqrt(2.0 / (9 * c))
    b = torch.randn(c, generator=g(7)) * 0.1
    gy = torch.randn(n, c, h, w, generator=g(8))
    xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.conv2d(xr, wr, br, padding=1)          # (no ReLU: no mask decisions that could differ between the two sides)
    yr.backward(gy)
    xd, wd, bd = x.to(DEV).requires_grad_(), wt.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    yd = ops.conv3x3(xd, wd, bd, False)
    close(yd, yr, 1e-4, 1e-4 * float(yr.abs().max()), "oversized map forward")
    yd.backward(gy.to(DEV))
    for name, a, r in (("dgrad", xd.grad, xr.grad), ("dW", wd.grad, wr.grad), ("db", bd.grad, br.grad)):
        close(a, r, 1e-4, 1e-4 * float(r.abs().max()), f"oversized map {name}")
    with pytest.raises(_lib.PtmiError):          # a Winograd pack handed to a map that is routed to the direct kernel is refused
        with torch.no_grad():
            wp = ops.conv3x3_pack(wd.detach(), 0, 1)
            ops.conv3x3_raw(xd.detach(), wp, bd.detach(), None, c, 1)
