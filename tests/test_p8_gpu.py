"""GPU parity tests (pytest -m gpu) of the bf16-STORAGE convolution path (csrc/p8.hip, C ABI ptmi_p8_*; SOLVER.AMP.ENABLED,
reference pt/engine/trainer.py:98, pt/modeling/backbone/vgg.py:45-72) through the C ABI.

The statement: P8 tensors hold bf16-rounded values; a convolution multiplies them exactly (bf16 x bf16 fits fp32) and accumulates
in fp32, so the fp32 accumulator equals torch CPU fp32 conv2d on the bf16-rounded operands up to summation order, and the stored
result is that value rounded to bf16: |a - b| <= 2^-7 |b| + 1e-4 max|b| (one bf16 ulp is 2^-8 relative; a rounding boundary may
be crossed).  Index-like outputs (pool argmax routing, ReLU masks, pad zeros) are exact."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def g(seed):
    return torch.Generator().manual_seed(seed)


def rb(t):
    """round to bf16 and back (nearest even)"""
    return t.to(torch.bfloat16).to(torch.float32)


def bf16_close(a, b, what, extra=0.0):
    a, b = a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy()
    assert a.shape == b.shape, f"{what}: {a.shape} vs {b.shape}"
    err = np.abs(a - b)
    tol = 2.0 ** -7 * np.abs(b) + (1e-4 + extra) * np.abs(b).max()
    assert (err <= tol).all(), (f"{what}: max abs err {err.max():.3e}, max |ref| {np.abs(b).max():.3e}, {int((err > tol).sum())} of "
                                f"{err.size} off, first at {np.argwhere(err > tol)[0]}")


def pads_are_zero(t, n, h, w):
    t = t.cpu()
    rows = torch.arange(t.shape[1])
    assert bool((t[:, rows % (h + 1) == 0] == 0).all()), "zero rows between images"
    assert bool((t[:, :, 0] == 0).all()), "zero column"


SHAPES = [  # n, cin, cout, h, w
    (2, 16, 64, 5, 7),          # one chunk, MT = 2 (64-channel tile, 32-row workgroup tile)
    (1, 32, 128, 19, 35),       # two chunks (register sets swap roles), MT = 4, two column tiles
    (2, 64, 64, 24, 40),        # images stacked in the row dimension: a tile spans two images
    (1, 128, 256, 13, 33),      # two channel tiles, eight chunks
    (3, 16, 8, 4, 3),           # fewer channels than a tile, narrower than a piece
    (1, 48, 72, 9, 83),         # odd chunk count (3), ragged channel tile
    (2, 64, 128, 50, 83),       # the block-5 map
    (1, 16, 24, 1, 1),          # a single pixel
]


@pytest.mark.parametrize("n,cin,cout,h,w", SHAPES)
def test_p8_roundtrip_and_conv_forward_dgrad(n, cin, cout, h, w):
    from probabilisticteacher_amd import p8
    x = rb(torch.randn(n, cin, h, w, generator=g(1)))
    wt = rb(torch.randn(cout, cin, 3, 3, generator=g(2)) * math.sqrt(2.0 / (9 * cin)))
    b = torch.randn(cout, generator=g(3)) * 0.1
    xp = p8.from_nchw(x.to(DEV))
    assert xp.shape == (2 * ((cin + 15) // 16), n * (h + 1) + 1, w + 1, 8)
    pads_are_zero(xp, n, h, w)
    assert torch.equal(p8.to_nchw(xp, n, cin, h, w).cpu(), x), "bf16-representable values survive the round trip exactly"
    # forward, both epilogues
    for epi in (0, 1):
        yr = F.conv2d(x, wt, b, padding=1)
        if epi == 1:
            yr = F.relu(yr)
        yp = p8.conv3x3_raw(xp, p8.pack_weights(wt.to(DEV), 0), b.to(DEV), None, n, cin, cout, h, w, epi)
        pads_are_zero(yp, n, h, w)
        bf16_close(p8.to_nchw(yp, n, cout, h, w), yr, f"forward epilogue {epi}")
    # dgrad: dX = conv(dY, W^T rotated)
    if True:
        gy = rb(torch.randn(n, cout, h, w, generator=g(4)))
        xr = x.clone().requires_grad_()
        F.conv2d(xr, wt, None, padding=1).backward(gy)
        gp = p8.from_nchw(gy.to(DEV))
        wpd = p8.pack_weights(wt.to(DEV), 1)
        dx = p8.conv3x3_raw(gp, wpd, None, None, n, cout, cin, h, w, 2)
        pads_are_zero(dx, n, h, w)
        bf16_close(p8.to_nchw(dx, n, cin, h, w), xr.grad, "dgrad")
        # epilogue 3: times the ReLU mask of the producer (a post-ReLU activation with exact zeros)
        act = rb(torch.relu(torch.randn(n, cin, h, w, generator=g(5))))
        dxm = p8.conv3x3_raw(gp, wpd, None, p8.from_nchw(act.to(DEV)), n, cout, cin, h, w, 3)
        pads_are_zero(dxm, n, h, w)
        bf16_close(p8.to_nchw(dxm, n, cin, h, w), xr.grad * (act > 0), "dgrad + mask")


def test_p8_stem_layer_three_channels_padded_to_sixteen():
    """conv1_1 (vgg.py:44: 3 -> 64): the image becomes a 16-channel P8 tensor (13 zero channels) and runs on the MFMA kernel"""
    from probabilisticteacher_amd import p8
    n, h, w = 2, 37, 53
    x = rb(torch.randn(n, 3, h, w, generator=g(7)) * 50)
    wt = rb(torch.randn(64, 3, 3, 3, generator=g(8)) * 0.2)
    b = torch.randn(64, generator=g(9))
    xp = p8.from_nchw(x.to(DEV))
    assert xp.shape[0] == 2 and bool((xp[1] == 0).all()) and bool((xp[0, :, :, 3:] == 0).all())
    yp = p8.conv3x3_raw(xp, p8.pack_weights(wt.to(DEV), 0), b.to(DEV), None, n, 3, 64, h, w, 1)
    bf16_close(p8.to_nchw(yp, n, 64, h, w), F.relu(F.conv2d(x, wt, b, padding=1)), "stem")


@pytest.mark.parametrize("n,c,h,w", [(2, 16, 8, 10), (1, 64, 7, 9), (3, 8, 2, 2), (2, 128, 50, 83)])
def test_p8_maxpool_forward_backward_exact(n, c, h, w):
    from probabilisticteacher_amd import p8
    x = rb(torch.relu(torch.randn(n, c, h, w, generator=g(11))))
    x[:, :, : h // 2 * 2 : 2, : w // 2 * 2 : 2] = x[:, :, 1 : h // 2 * 2 : 2, : w // 2 * 2 : 2]        # ties: first maximum wins
    xr = x.clone().requires_grad_()
    yr = F.max_pool2d(xr, 2, 2)
    gy = rb(torch.randn(yr.shape, generator=g(12)))
    yr.backward(gy)
    xp = p8.from_nchw(x.to(DEV))
    yp = p8.maxpool_fwd(xp, n, c, h, w)
    pads_are_zero(yp, n, h // 2, w // 2)
    assert torch.equal(p8.to_nchw(yp, n, c, h // 2, w // 2).cpu(), yr.detach())
    gp = p8.from_nchw(gy.to(DEV))
    for relu_mask in (False, True):
        dx = p8.maxpool_bwd(xp, gp, n, c, h, w, relu_mask)
        pads_are_zero(dx, n, h, w)
        want = xr.grad * (x > 0) if relu_mask else xr.grad
        assert torch.equal(p8.to_nchw(dx, n, c, h, w).cpu(), want), f"pool backward (relu_mask={relu_mask})"
    # relu_bwd
    y = rb(torch.randn(n, c, h, w, generator=g(13)))
    y[0, 0, 0, 0] = -0.0
    d = rb(torch.randn(n, c, h, w, generator=g(14)))
    dz = p8.relu_bwd(p8.from_nchw(d.to(DEV)), p8.from_nchw(y.to(DEV)))
    assert torch.equal(p8.to_nchw(dz, n, c, h, w).cpu(), d * (y > 0))


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 16, 64, 8, 10), (3, 32, 128, 33, 37), (1, 64, 48, 7, 9), (2, 16, 200, 2, 3),
                                            (5, 24, 64, 40, 70), (2, 64, 64, 800, 1333), (2, 128, 128, 400, 666)])
def test_p8_conv_with_pooling_epilogue_equals_conv_then_pool(n, cin, cout, h, w):
    """epilogue 4 (bias + ReLU + 2x2 max-pool in the conv kernel's epilogue, blocks without a backward pass): bit for bit the
    pooled tensor that the ReLU launch followed by the pool kernel writes, pads included -- odd heights / widths (floor),
    both channel-tile widths, several images (the tile grid restarts per image), the two frozen layers at 1333 x 800."""
    from probabilisticteacher_amd import p8
    x = rb(torch.randn(n, cin, h, w, generator=g(31)))
    wt = rb(torch.randn(cout, cin, 3, 3, generator=g(32)) * (2.0 / (9 * cin)) ** 0.5)
    b = torch.randn(cout, generator=g(33)) * 0.1
    xp = p8.from_nchw(x.to(DEV))
    wp = p8.pack_weights(wt.to(DEV), 0)
    full = p8.conv3x3_raw(xp, wp, b.to(DEV), None, n, cin, cout, h, w, 1)
    want = p8.maxpool_fwd(full, n, cout, h, w)
    got = torch.full_like(want, float("nan"))
    from probabilisticteacher_amd import _lib, ops
    _lib.call("ptmi_p8_conv3x3", ops._ptr(xp), ops._ptr(wp), ops._ptr(b.to(DEV)), None, ops._ptr(got), n, cin, cout, h, w, 4, ops._stream())
    pads_are_zero(got, n, h // 2, w // 2)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert torch.equal(p8.conv3x3_raw(xp, wp, b.to(DEV), None, n, cin, cout, h, w, 4).view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("name,cin,cout,h,w", [("conv1_2", 64, 64, 800, 1333), ("conv2_2", 128, 128, 400, 666),
                                               ("conv3_2", 256, 256, 200, 333), ("conv4_2", 512, 512, 100, 166),
                                               ("conv5_2", 512, 512, 50, 83)])
def test_p8_conv_full_size_layers(name, cin, cout, h, w):
    """the 1333 x 800 layer shapes (n = 2): forward + ReLU and masked dgrad against torch CPU fp32 on the rounded operands"""
    import os
    from probabilisticteacher_amd import p8
    torch.set_num_threads(max(2, min(os.cpu_count() or 2, 64)))
    n = 2
    x = rb(torch.relu(torch.randn(n, cin, h, w, generator=g(21))))
    wt = rb(torch.randn(cout, cin, 3, 3, generator=g(22)) * math.sqrt(2.0 / (9 * cin)))
    b = torch.randn(cout, generator=g(23)) * 0.1
    xp = p8.from_nchw(x.to(DEV))
    yp = p8.conv3x3_raw(xp, p8.pack_weights(wt.to(DEV), 0), b.to(DEV), None, n, cin, cout, h, w, 1)
    bf16_close(p8.to_nchw(yp, n, cout, h, w), F.relu(F.conv2d(x, wt, b, padding=1)), f"{name} forward")
    gy = rb(torch.randn(n, cout, h, w, generator=g(24)))
    xr = x.clone().requires_grad_()
    F.conv2d(xr, wt, None, padding=1).backward(gy)
    dx = p8.conv3x3_raw(p8.from_nchw(gy.to(DEV)), p8.pack_weights(wt.to(DEV), 1), None, xp, n, cout, cin, h, w, 3)
    bf16_close(p8.to_nchw(dx, n, cin, h, w), xr.grad * (x > 0), f"{name} dgrad + mask")


WG_SHAPES = [  # n, cin, cout, h, w
    (2, 64, 128, 5, 7),         # one channel-tile pair, fewer K tiles than workgroup slots
    (1, 64, 64, 19, 35),        # half a co tile; two K-tile columns, the second with 4 valid columns (k-step skipped)
    (2, 128, 256, 24, 40),      # 2 x 2 channel tiles; images stacked in the K range
    (3, 72, 40, 4, 3),          # ragged channel tiles both ways
    (1, 256, 128, 50, 83),      # the block-5 map, four ci tiles
    (1, 16, 8, 1, 1),
]


@pytest.mark.parametrize("n,cin,cout,h,w", WG_SHAPES)
def test_p8_wgrad_and_bias_grad(n, cin, cout, h, w):
    from probabilisticteacher_amd import p8
    x = rb(torch.relu(torch.randn(n, cin, h, w, generator=g(31))))
    gy = rb(torch.randn(n, cout, h, w, generator=g(32)))
    wr = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    br = torch.zeros(cout, requires_grad=True)
    F.conv2d(x, wr, br, padding=1).backward(gy)
    dw, db = p8.wgrad(p8.from_nchw(x.to(DEV)), p8.from_nchw(gy.to(DEV)), n, cin, cout, h, w)
    # fp32 accumulation of exact bf16 products: the fp32 bar (summation order only)
    s = float(wr.grad.abs().max())
    err = float((dw.cpu() - wr.grad).abs().max())
    assert err <= 1e-4 * s + 1e-6, f"dW: max abs err {err:.3e} vs scale {s:.3e}"
    sb = float(br.grad.abs().max())
    errb = float((db.cpu() - br.grad).abs().max())
    assert errb <= 1e-4 * sb + 1e-5, f"db: max abs err {errb:.3e} vs scale {sb:.3e}"
    dw2, db2 = p8.wgrad(p8.from_nchw(x.to(DEV)), p8.from_nchw(gy.to(DEV)), n, cin, cout, h, w)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "fixed-order split reduction: bitwise reproducible"


@pytest.mark.parametrize("name,cin,cout,h,w", [("conv3_2", 256, 256, 200, 333), ("conv4_2", 512, 512, 100, 166),
                                               ("conv5_2", 512, 512, 50, 83), ("conv3_1", 128, 256, 200, 333)])
def test_p8_wgrad_full_size_layers(name, cin, cout, h, w):
    import os
    from probabilisticteacher_amd import p8
    torch.set_num_threads(max(2, min(os.cpu_count() or 2, 64)))
    n = 2
    x = rb(torch.relu(torch.randn(n, cin, h, w, generator=g(41))))
    gy = rb(torch.randn(n, cout, h, w, generator=g(42)))
    wr = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    br = torch.zeros(cout, requires_grad=True)
    F.conv2d(x, wr, br, padding=1).backward(gy)
    dw, db = p8.wgrad(p8.from_nchw(x.to(DEV)), p8.from_nchw(gy.to(DEV)), n, cin, cout, h, w)
    s, sb = float(wr.grad.abs().max()), float(br.grad.abs().max())
    assert float((dw.cpu() - wr.grad).abs().max()) <= 1e-4 * s, f"{name} dW"
    assert float((db.cpu() - br.grad).abs().max()) <= 1e-4 * sb, f"{name} db"


@pytest.mark.parametrize("waves", [2, 4, 16, 64])      # weight gradient: min(waves, 16); 2 = PTrainer.ddp_wgrad_waves(True)
def test_p8_conv_and_wgrad_waves_entry_points(waves):
    """round 6: ptmi_p8_conv3x3_waves -- more, shorter persistent workgroups (what PTrainer selects under DDP with SOLVER.AMP.ENABLED) --
    gives the one-fill kernel's output BIT FOR BIT (a tile's arithmetic does not depend on which workgroup computes it), every epilogue;
    ptmi_p8_wgrad_waves agrees with one fill up to the summation order of the split partials and repeats bitwise"""
    from probabilisticteacher_amd import ops, p8
    n, cin, cout, h, w = 3, 64, 128, 45, 70
    x = rb(torch.relu(torch.randn(n, cin, h, w, generator=g(81))))
    wt = rb(torch.randn(cout, cin, 3, 3, generator=g(82)) * 0.05)
    b = torch.randn(cout, generator=g(83)) * 0.1
    gy = rb(torch.randn(n, cout, h, w, generator=g(84)))
    xp, gp = p8.from_nchw(x.to(DEV)), p8.from_nchw(gy.to(DEV))
    wp = p8.pack_weights(wt.to(DEV), 0)
    mask = p8.from_nchw(rb(torch.randn(n, cout, h, w, generator=g(85))).to(DEV))
    outs = {}
    try:
        for wv in (1, waves):
            ops.set_p8_conv_waves(wv)
            ops.set_wgrad_waves(min(wv, 16))
            outs[wv] = [p8.conv3x3_raw(xp, wp, b.to(DEV) if epi in (0, 1, 4) else None, mask if epi == 3 else None, n, cin, cout, h, w, epi)
                        for epi in (0, 1, 2, 3, 4)] + list(p8.wgrad(xp, gp, n, cin, cout, h, w))
        again = list(p8.wgrad(xp, gp, n, cin, cout, h, w))
    finally:
        ops.set_p8_conv_waves(1)
        ops.set_wgrad_waves(1)
    for epi, (a, c) in enumerate(zip(outs[1][:5], outs[waves][:5])):
        assert torch.equal(a, c), f"epilogue {epi}: {waves} waves differ from one fill"
    for name, a, c, r in (("dW", outs[1][5], outs[waves][5], again[0]), ("db", outs[1][6], outs[waves][6], again[1])):
        assert torch.equal(c, r), f"{name}: not bitwise repeatable at {waves} waves"
        sc = float(a.abs().max())
        assert float((a - c).abs().max()) <= 1e-4 * sc, f"{name}: {waves} waves vs one fill"


def test_p8_wgrad_oversize_fallback_matches_native_kernel(monkeypatch):
    """p8.wgrad's route for a P8 tensor beyond the kernel's 32-bit offsets (ptmi_p8_wgrad_fits = 0) -- the direct fp32
    split-K kernel on widened operands, over GROUPS of images with accumulate = 1 -- forced on a small shape and compared with the
    native bf16 kernel (same exact bf16 products, fp32 accumulation in another order: the fp32 bar) and with torch CPU."""
    from probabilisticteacher_amd import _lib, ops, p8
    n, cin, cout, h, w = 5, 32, 48, 21, 30
    x = rb(torch.relu(torch.randn(n, cin, h, w, generator=g(61))))
    gy = rb(torch.randn(n, cout, h, w, generator=g(62)))
    wr = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    br = torch.zeros(cout, requires_grad=True)
    F.conv2d(x, wr, br, padding=1).backward(gy)
    xp, gp = p8.from_nchw(x.to(DEV)), p8.from_nchw(gy.to(DEV))
    dw_native, db_native = p8.wgrad(xp, gp, n, cin, cout, h, w)
    lib = _lib.load()

    class NoFit:
        def __getattr__(self, name):
            return (lambda *a: 0) if name == "ptmi_p8_wgrad_fits" else getattr(lib, name)
    monkeypatch.setattr(_lib, "load", lambda: NoFit())
    monkeypatch.setattr(p8, "_WGRAD_FALLBACK_BYTES", 2 * 4 * max(cin, cout) * h * w)      # groups of 2 images: 2 + 2 + 1
    ops.profile_start()
    dw, db = p8.wgrad(xp, gp, n, cin, cout, h, w)
    prof = ops.profile_stop()
    assert list(prof) == ["conv3x3_wgrad"] and prof["conv3x3_wgrad"]["calls"] == 3, prof
    s, sb = float(wr.grad.abs().max()), float(br.grad.abs().max())
    for name, a, b, sc in (("dW vs native", dw, dw_native, s), ("db vs native", db, db_native, sb),
                           ("dW vs torch", dw.cpu(), wr.grad, s), ("db vs torch", db.cpu(), br.grad, sb)):
        err = float((a - b).abs().max())
        assert err <= 1e-4 * sc + 1e-6, f"{name}: max abs err {err:.3e} vs scale {sc:.3e}"


def test_p8_gemm_nt_rejects_operands_beyond_32_bit_offsets():
    """p8.gemm_nt raises PtmiError (no silent wrap) when a packed operand exceeds the kernel's 4 GiB buffer range;
    nothing is launched, so the operands can be tiny stand-ins"""
    from probabilisticteacher_amd import _lib, p8
    m, n, k = 90000, 1024, 25088                    # fc1 with 90 k ROIs: 4.5 GB packed
    assert not _lib.load().ptmi_p8_gemm_nt_fits(m, n, k) and _lib.load().ptmi_p8_gemm_nt_fits(32000, n, k)
    a = torch.zeros(8, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_lib.PtmiError, match="32-bit buffer offsets"):
        p8.gemm_nt(a, a, m, n, k)


GEMM_SHAPES = [  # m (rows), n (outputs), k
    (300, 70, 200),            # one ragged tile, k not a multiple of 64 (zero-filled octets), no split
    (512, 256, 1024),          # exact tiles
    (257, 513, 136),           # one row / column beyond a tile boundary
    (64, 1024, 25088),         # few tiles, long k: split-K through the workspace
    (1000, 9, 1024),           # narrow output (predictor-like)
    (5, 3, 8),
]


@pytest.mark.parametrize("m,n,k", GEMM_SHAPES)
def test_p8_gemm_nt_and_packs(m, n, k):
    """ptmi_p8m_pack (both source orientations) + ptmi_p8_gemm_nt against float64 products of the bf16-rounded operands"""
    from probabilisticteacher_amd import p8
    a = rb(torch.randn(m, k, generator=g(51)))
    b = rb(torch.randn(n, k, generator=g(52)))
    bias = torch.randn(n, generator=g(53))
    ref = a.double() @ b.double().t()
    ap, bp = p8.pack_matrix(a.to(DEV), m, k, k, True), p8.pack_matrix(b.to(DEV), n, k, k, True)
    assert ap.shape == ((k + 7) // 8, m, 8)
    # the pack itself: element (row, kk) of the packed operand
    back = ap.cpu().float().permute(1, 0, 2).reshape(m, -1)[:, :k]
    assert torch.equal(back, a), "k-major pack"
    at = p8.pack_matrix(a.t().contiguous().to(DEV), m, k, m, False)            # the same operand from the transposed source
    assert torch.equal(at, ap), "row-major pack of the transposed source gives the same operand"
    for relu in (False, True):
        c = p8.gemm_nt(ap, bp, m, n, k, bias.to(DEV), relu).cpu().double()
        want = ref + bias.double()
        if relu:
            want = want.clamp(min=0)
        err = float((c - want).abs().max())
        assert err <= 1e-5 * float(want.abs().max()) + 4e-6 * math.sqrt(k) + 1e-5, f"gemm relu={relu}: max abs err {err:.3e}"
    cs = p8.gemm_nt(ap, bp, m, n, k, None, False, swapped=True).cpu().double()       # C^T = B . A^T stored transposed: the same matrix
    assert float((cs - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 4e-6 * math.sqrt(k) + 1e-5, "swapped / transposed-store form"
    c1 = p8.gemm_nt(ap, bp, m, n, k, None, False)
    assert torch.equal(c1, p8.gemm_nt(ap, bp, m, n, k, None, False)), "bitwise reproducible (fixed-order split reduction)"


def test_p8_linear_autograd_matches_torch_on_rounded_operands():
    """p8._LinearP8 (what ops.linear runs for fc1 in "bf16" mode): y, dX, dW, db against torch CPU on bf16-rounded operands"""
    from probabilisticteacher_amd import ops, p8
    r, k, n = 700, 4608, 320
    x = torch.randn(r, k, generator=g(61))
    w = torch.randn(n, k, generator=g(62)) * 0.02
    b = torch.randn(n, generator=g(63))
    gy = torch.randn(r, n, generator=g(64))
    xr, wr, br = rb(x).requires_grad_(), rb(w).requires_grad_(), b.clone().requires_grad_()
    zr = F.linear(xr, wr, br)
    yr = F.relu(zr)
    yr.backward(rb(gy * (zr.detach() > 0)))          # the gradient that reaches the GEMMs is rounded
    ops.set_operand_rounding("bf16")
    try:
        xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, w, b))
        yd = ops.linear(xd, wd, bd, True)
        assert isinstance(yd.grad_fn, p8._LinearP8._backward_cls) or "LinearP8" in type(yd.grad_fn).__name__
        yd.backward(gy.to(DEV))
    finally:
        ops.set_operand_rounding(None)
    sc = lambda t: float(t.abs().max())
    assert float((yd.detach().cpu() - yr.detach()).abs().max()) <= 1e-4 * sc(yr) + 1e-4
    stable = zr.detach().abs() > 1e-4
    assert float(stable.float().mean()) > 0.999
    if bool(stable.all()):
        assert float((xd.grad.cpu() - xr.grad).abs().max()) <= 1e-4 * sc(xr.grad) + 1e-5
        assert float((wd.grad.cpu() - wr.grad).abs().max()) <= 2e-4 * sc(wr.grad) + 1e-5
        assert float((bd.grad.cpu() - br.grad).abs().max()) <= 2e-4 * sc(br.grad) + 1e-4


def test_p8_conv3_2_at_n48_bench_launch_shape_vs_torch():
    """ONE launch of each storage kernel at the joint student pass's batch (n = 48, 256 -> 256 at 200 x 333; `bench.py --amp` runs
    this shape): forward + ReLU, masked dgrad (persistent workgroups walking ~17 tiles each, images stacked in the row dimension),
    weight + bias gradient (32 K splits) -- against torch CPU fp32 on the bf16-rounded operands"""
    import os
    from probabilisticteacher_amd import p8
    torch.set_num_threads(max(2, min(os.cpu_count() or 2, 64)))
    n, c, h, w = 48, 256, 200, 333
    x = rb(torch.relu(torch.randn(n, c, h, w, generator=g(71))))
    wt = rb(torch.randn(c, c, 3, 3, generator=g(72)) * math.sqrt(2.0 / (9 * c)))
    b = torch.randn(c, generator=g(73)) * 0.1
    gy = rb(torch.randn(n, c, h, w, generator=g(74)))
    xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.conv2d(xr, wr, br, padding=1)
    yr.backward(gy)
    xp = p8.from_nchw(x.to(DEV))
    yp = p8.conv3x3_raw(xp, p8.pack_weights(wt.to(DEV), 0), b.to(DEV), None, n, c, c, h, w, 1)
    bf16_close(p8.to_nchw(yp, n, c, h, w), F.relu(yr.detach()), "conv3_2 forward n=48")
    del yp
    gp = p8.from_nchw(gy.to(DEV))
    dx = p8.conv3x3_raw(gp, p8.pack_weights(wt.to(DEV), 1), None, xp, n, c, c, h, w, 3)
    bf16_close(p8.to_nchw(dx, n, c, h, w), xr.grad * (x > 0), "conv3_2 dgrad + mask n=48")
    del dx
    dw, db = p8.wgrad(xp, gp, n, c, c, h, w)
    s, sb = float(wr.grad.abs().max()), float(br.grad.abs().max())
    assert float((dw.cpu() - wr.grad).abs().max()) <= 1e-4 * s, "conv3_2 wgrad n=48"
    assert float((db.cpu() - br.grad).abs().max()) <= 1e-4 * sb, "conv3_2 bias grad n=48"


@pytest.mark.parametrize("n,c,h,w,per", [(2, 16, 25, 31, 19), (3, 64, 50, 83, 171), (1, 8, 12, 9, 8)])
def test_roi_align_writing_the_linear_layers_bf16_operands_equals_pool_then_pack(n, c, h, w, per):
    """SOLVER.AMP.ENABLED box head: ROIAlign -> flatten -> Linear + ReLU as one node whose ROIAlign kernel writes the GEMM's bf16
    operands (xk, and xt for the weight gradient) itself == the fp32 ROIAlign followed by p8.linear's operand packs: the same bf16
    values, hence y, dW, db and the feature-map gradient bit for bit (ROI counts that are not multiples of 8: the zero tail of xt)"""
    from probabilisticteacher_amd import ops, p8
    gen = g(n * 1000 + c)
    feat = torch.randn(n, c, h, w, generator=gen)
    r = n * per
    ctr = torch.rand(r, 2, generator=gen) * torch.tensor([w * 16.0, h * 16.0])
    wh = 24 + torch.rand(r, 2, generator=gen) * 200
    img = torch.arange(n).repeat_interleave(per).float()
    rois = torch.cat([img[:, None], ctr - wh / 2, ctr + wh / 2], 1).to(DEV)
    offs = torch.arange(0, (n + 1) * per, per, dtype=torch.int32, device=DEV)
    k, nout = c * 49, 40
    wt = (torch.randn(nout, k, generator=gen) * 0.05).to(DEV)
    b = (torch.randn(nout, generator=gen) * 0.1).to(DEV)
    gy = torch.randn(r, nout, generator=gen).to(DEV)
    assert ops.roi_align_p8m_fits(c, h, w, 7)
    # the operands themselves
    xk, xt = ops.roi_align_p8m(feat.to(DEV), rois, offs, 7, 1 / 16, True)
    x32 = ops.roi_align(feat.to(DEV), rois, 7, 1 / 16, offs).flatten(1)
    assert torch.equal(xk.view(torch.int16), p8.pack_matrix(x32, r, k, k, True).view(torch.int16)), "xk"
    assert torch.equal(xt.view(torch.int16), p8.pack_matrix(x32, k, r, k, False).view(torch.int16)), "xt (incl. its zero tail)"
    res = []
    for fused in (True, False):
        f = feat.to(DEV).requires_grad_()
        w_, b_ = wt.clone().requires_grad_(), b.clone().requires_grad_()
        if fused:
            y = p8.roi_align_linear(f, rois, offs, 7, 1 / 16, w_, b_, True)
        else:
            y = p8.linear(ops.roi_align(f, rois, 7, 1 / 16, offs).flatten(1), w_, b_, True)
        y.backward(gy)
        res.append((y.detach(), f.grad, w_.grad, b_.grad))
    for a, bb, what in zip(res[0], res[1], ("y", "d feature map", "dW", "db")):
        assert torch.equal(a, bb), what
