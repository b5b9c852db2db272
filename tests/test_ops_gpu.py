"""GPU parity tests (pytest -m gpu): every HIP operator of libptmi355.so against the CPU oracle / stock
torch CPU fp32 ops on identical seeded inputs.

Tolerances (stated per test): integer / index / mask outputs bit-exact; fp32 GEMM-like outputs
rtol 1e-4 (different accumulation order than oneDNN), elementwise fp32 rtol 1e-5."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import d2, pt as opt
from tests.helpers import load, records

pytestmark = pytest.mark.gpu


def _raw():
    """(call, ptr, stream) for tests that drive a C-ABI entry point directly"""
    from probabilisticteacher_amd import _lib, ops as _ops
    return _lib.call, _ops._ptr, _ops._stream


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from probabilisticteacher_amd import ops as _ops
    return _ops


DEV = "cuda:0"


def g(seed):
    return torch.Generator().manual_seed(seed)


def close(a, b, rtol, atol, what=""):
    a = a.detach().cpu().double().numpy()
    b = b.detach().cpu().double().numpy()
    assert a.shape == b.shape, f"{what}: {a.shape} vs {b.shape}"
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), f"{what}: max abs err {err.max():.3e}, max |ref| {np.abs(b).max():.3e}"


# ------------------------------------------------------------------------------------------ conv
@pytest.mark.parametrize("n,cin,cout,h,w,relu", [
    (2, 3, 64, 37, 45, True),       # conv1_1 path (CK=4, BM=64), ragged tiles
    (1, 64, 64, 24, 40, True),      # BM=64, CK=8
    (2, 64, 128, 19, 35, True),     # BM=128
    (1, 128, 256, 13, 33, False),   # no relu
    (1, 256, 512, 9, 83, True),     # W=83 as in the 1333x800 block5 map
    (1, 20, 70, 11, 17, True),      # channel counts that are not multiples of the tiles
    (1, 32, 128, 5, 166, True),     # W=166 / 333: right-edge tiles whose 16-B pieces straddle the image edge
    (2, 32, 128, 3, 333, False),
    (1, 32, 128, 7, 32, True),      # exactly one tile per row
    (1, 32, 128, 6, 36, True),      # last tile 4 wide
    (3, 32, 64, 4, 3, True),        # narrower than one 16-B piece
    (1, 40, 130, 1, 70, True),      # a single row: top and bottom halo at once
    (1, 16, 64, 9, 701, True),      # W >= 600: 1 x 32 pixel blocks (row-major variant), odd height, ragged right edge
    (2, 32, 130, 5, 640, False),    # ... 128-channel tiles, exact tile columns, partial channel tile
])
def test_conv3x3_fwd_bwd(ops, n, cin, cout, h, w, relu):
    gen = g(n * 1000 + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.conv2d(xr, wr, br, padding=1)
    if relu:
        yr = F.relu(yr)
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)

    xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, wt, b))
    yd = ops.conv3x3(xd, wd, bd, relu)
    close(yd, yr, 1e-4, 1e-4, "conv fwd")
    yd.backward(gy.to(DEV))
    close(xd.grad, xr.grad, 1e-4, 2e-4, "conv dgrad")
    close(wd.grad, wr.grad, 1e-4, 1e-3, "conv wgrad")
    close(bd.grad, br.grad, 1e-4, 1e-3, "conv bias grad")


def test_conv3x3_linearity_full_size(ops):
    """Size-independent property at a BASELINE-sized layer: conv(a*x1 + x2) == a*conv(x1) + conv(x2) (no relu)."""
    gen = g(5)
    cin, cout, h, w = 64, 64, 200, 333
    x1 = torch.randn(1, cin, h, w, generator=gen).to(DEV)
    x2 = torch.randn(1, cin, h, w, generator=gen).to(DEV)
    wt = (torch.randn(cout, cin, 3, 3, generator=gen) * 0.05).to(DEV)
    zero = torch.zeros(cout, device=DEV)
    lhs = ops.conv3x3(2.5 * x1 + x2, wt, zero, False)
    rhs = 2.5 * ops.conv3x3(x1, wt, zero, False) + ops.conv3x3(x2, wt, zero, False)
    close(lhs, rhs, 1e-4, 1e-3, "linearity")
    # and a sampled direct check against torch CPU on a crop that includes image borders
    ref = F.conv2d(x1[:, :, :40, :50].cpu(), wt.cpu(), None, padding=1)
    got = ops.conv3x3(x1, wt, zero, False)[:, :, :39, :49].cpu()
    close(got, ref[:, :, :39, :49], 1e-4, 1e-4, "crop")


def test_maxpool(ops):
    gen = g(7)
    x = torch.randn(3, 5, 21, 27, generator=gen)
    x[0, 0, 0:2, 0:2] = 1.0      # tie inside a window -> first max gets the gradient
    xr = x.clone().requires_grad_()
    yr = F.max_pool2d(xr, 2, 2)
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xd = x.to(DEV).requires_grad_()
    yd = ops.maxpool2x2(xd)
    assert torch.equal(yd.cpu(), yr.detach()), "maxpool fwd must be bit-exact"
    yd.backward(gy.to(DEV))
    assert torch.equal(xd.grad.cpu(), xr.grad), "maxpool bwd must be bit-exact"


# ------------------------------------------------------------------------------------------ gemm family
@pytest.mark.parametrize("r,k,nout,relu", [(37, 200, 70, True), (300, 1000, 130, False), (0, 64, 16, True),
                                           (129, 33, 257, True)])
def test_linear(ops, r, k, nout, relu):
    gen = g(r + k + nout)
    x = torch.randn(r, k, generator=gen)
    w = torch.randn(nout, k, generator=gen) / math.sqrt(k)
    b = torch.randn(nout, generator=gen)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.linear(xr, wr, br)
    if relu:
        yr = F.relu(yr)
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, w, b))
    yd = ops.linear(xd, wd, bd, relu)
    close(yd, yr, 1e-4, 1e-4, "linear fwd")
    yd.backward(gy.to(DEV))
    close(xd.grad, xr.grad, 1e-4, 1e-4, "linear dx")
    close(wd.grad, wr.grad, 1e-4, 1e-3, "linear dw")
    close(bd.grad, br.grad, 1e-4, 1e-3, "linear db")


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(130, 66, 70), (64, 257, 31), (5, 3, 2), (256, 128, 96)])
def test_gemm_all_layouts(ops, ta, tb, m, n, k):
    """ptmi_gemm_f32 in all four storage combinations, ragged M / N / K (K % 4 != 0 exercises the straddling 16-B
    pieces), batched with strides, against a float64 matmul."""
    gen = g(m * 7 + n * 3 + k + ta * 2 + tb)
    batch = 2
    a = torch.randn(batch, *((k, m) if ta else (m, k)), generator=gen)
    b = torch.randn(batch, *((n, k) if tb else (k, n)), generator=gen)
    am = a.transpose(1, 2) if ta else a
    bm = b.transpose(1, 2) if tb else b
    ref = torch.bmm(am.double(), bm.double()).float()
    ad, bd = a.to(DEV).contiguous(), b.to(DEV).contiguous()
    out = ops.gemm(ad, bd, m, n, k, a.shape[2], b.shape[2], ta, tb, batch=batch, stride_a=a.shape[1] * a.shape[2],
                   stride_b=b.shape[1] * b.shape[2], stride_c=m * n)
    close(out, ref, 1e-4, 1e-4, f"gemm ta={ta} tb={tb}")
    out2 = ops.gemm(ad, bd, m, n, k, a.shape[2], b.shape[2], ta, tb, batch=batch, stride_a=a.shape[1] * a.shape[2],
                    stride_b=b.shape[1] * b.shape[2], stride_c=m * n)
    assert torch.equal(out, out2), "gemm must be bitwise reproducible"


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(200, 300, 5000), (9, 130, 4099), (1024, 1024, 2048 + 13)])
def test_gemm_split_k(ops, ta, tb, m, n, k):
    """shapes for which ptmi_gemm_ws_floats() asks for a workspace run split-K (few tiles, long K: the box head's weight
    gradients): all four layouts, ragged K slices, bias / ReLU / accumulate applied by the fixed-order reduction, bitwise
    reproducible, and equal (to rounding) to the unsplit kernel that runs when no workspace is passed"""
    from probabilisticteacher_amd import _lib
    nws = _lib.load().ptmi_gemm_ws_floats(m, n, k, 1)
    assert nws >= 2 * m * n, "this shape is meant to split"
    gen = g(m + n + k + 2 * ta + tb)
    a = torch.randn(*((k, m) if ta else (m, k)), generator=gen)
    b = torch.randn(*((n, k) if tb else (k, n)), generator=gen)
    bias = torch.randn(n, generator=gen)
    c0 = torch.randn(m, n, generator=gen)
    am, bm = (a.t() if ta else a), (b.t() if tb else b)
    ref = torch.relu(am.double() @ bm.double() + bias.double() + c0.double()).float()
    ad, bd, biasd = a.to(DEV).contiguous(), b.to(DEV).contiguous(), bias.to(DEV)
    out = ops.gemm(ad, bd, m, n, k, a.shape[1], b.shape[1], ta, tb, bias=biasd, bias_mode=2, relu=True, out=c0.to(DEV), accumulate=True)
    close(out, ref, 1e-4, 2e-4 * math.sqrt(k / 1000.0), f"split-K gemm ta={ta} tb={tb}")
    out2 = ops.gemm(ad, bd, m, n, k, a.shape[1], b.shape[1], ta, tb, bias=biasd, bias_mode=2, relu=True, out=c0.to(DEV), accumulate=True)
    assert torch.equal(out, out2), "split-K gemm must be bitwise reproducible"
    # the same call without a workspace: plain kernel
    call, _p, _stream = _raw()
    plain = c0.to(DEV)
    call("ptmi_gemm_f32", _p(ad), _p(bd), _p(plain), _p(biasd), m, n, k, a.shape[1], b.shape[1], n, ta, tb, 2, 1, 1, 1, 0, 0, 0,
         None, 0, _stream())
    close(plain, ref, 1e-4, 2e-4 * math.sqrt(k / 1000.0), "unsplit gemm")
    close(out, plain, 1e-5, 2e-4 * math.sqrt(k / 1000.0), "split vs unsplit (summation order differs)")


@pytest.mark.parametrize("rows,cols", [(199200, 15), (199200, 60), (5000, 7), (4095, 60), (70000, 1024), (3, 5), (0, 4)])
def test_colsum_tall_and_plain(ops, rows, cols):
    """ptmi_colsum_ws: tall matrices (the RPN 1x1 heads' bias gradients) are summed by row ranges and reduced in a fixed
    order; short / wide ones take the one-kernel path.  fp64 reference, bitwise reproducible, accumulate honoured."""
    from probabilisticteacher_amd import _lib
    gen = g(rows + cols)
    a = torch.randn(rows, cols, generator=gen)
    ref = a.double().sum(0)
    ad = a.to(DEV)
    out = ops.colsum(ad)
    close(out, ref.float(), 1e-5, 1e-5 * math.sqrt(max(rows, 1)), "colsum")
    assert torch.equal(out, ops.colsum(ad))
    nws = _lib.load().ptmi_colsum_ws_floats(rows, cols)
    assert (nws > 0) == (rows >= 4096 and cols < 2048)
    if rows:
        call, _p, _stream = _raw()
        acc = torch.ones(cols, device=DEV)
        ws = torch.empty(max(nws, 1), device=DEV)
        call("ptmi_colsum_ws", _p(ad), _p(acc), _p(ws), rows, cols, 1, _stream())
        close(acc, (ref + 1).float(), 1e-5, 1e-5 * math.sqrt(rows), "colsum accumulate")
        plain = torch.empty(cols, device=DEV)
        call("ptmi_colsum_ws", _p(ad), _p(plain), None, rows, cols, 0, _stream())          # no workspace: one-kernel path
        close(plain, ref.float(), 1e-5, 1e-5 * math.sqrt(rows), "colsum without workspace")


@pytest.mark.parametrize("m,n,k,batch", [(130, 200, 96, 1), (257, 72, 512, 3), (9, 130, 4099, 1), (200, 300, 5000, 1)])
def test_gemm_never_writes_outside_its_output(ops, m, n, k, batch):
    """Memory-safety canary for the buffer-store epilogue: the kernel advances rows through the store's scalar offset and
    relies on the buffer descriptor's range check to drop rows >= M of the last (ragged) tile.  The output sits inside a
    larger buffer filled with a sentinel: guard bands before it, after it and BETWEEN the batch items (stride_c > m * n) must
    come back untouched -- single, batched and split-K launches, M % 128 != 0."""
    gen = g(m + n + k)
    a = torch.randn(batch, m, k, generator=gen)
    b = torch.randn(batch, k, n, generator=gen)
    ref = torch.bmm(a.double(), b.double()).float()
    pad = 4096
    stride_c = m * n + pad
    buf = torch.full((pad + batch * stride_c + pad,), -12345.0, device=DEV)
    out = buf[pad:pad + batch * stride_c].view(batch, stride_c)[:, : m * n]
    ops.gemm(a.to(DEV).contiguous(), b.to(DEV).contiguous(), m, n, k, k, n, 0, 0, out=buf[pad:], batch=batch, stride_a=m * k,
             stride_b=k * n, stride_c=stride_c)
    close(out.reshape(batch, m, n), ref, 1e-4, 2e-4 * math.sqrt(max(k, 1000) / 1000.0), "gemm into a guarded buffer")
    guard = torch.ones_like(buf, dtype=torch.bool)
    for i in range(batch):
        guard[pad + i * stride_c: pad + i * stride_c + m * n] = False
    assert bool((buf[guard] == -12345.0).all()), f"{int((buf[guard] != -12345.0).sum())} guard elements were overwritten"


def test_rpn_head_flat_layout_equals_conv1x1_plus_permutes(ops):
    """ops.rpn_head_1x1 (the RPN head's two 1x1 convolutions as transposed GEMMs writing (N, HWA) / (N, HWA, 8) directly) against
    conv1x1 + the reference's permutes (rpn.py:97-113) and against torch CPU: forward and all five gradients."""
    gen = g(11)
    n, ci, h, w, a = 2, 64, 13, 21, 9
    x = torch.randn(n, ci, h, w, generator=gen)
    wo, bo = torch.randn(a, ci, 1, 1, generator=gen) * 0.1, torch.randn(a, generator=gen) * 0.1
    wd, bd = torch.randn(8 * a, ci, 1, 1, generator=gen) * 0.1, torch.randn(8 * a, generator=gen) * 0.1
    ref_in = [t.clone().requires_grad_() for t in (x, wo, bo, wd, bd)]
    lo = F.conv2d(ref_in[0], ref_in[1], ref_in[2]).permute(0, 2, 3, 1).reshape(n, -1)
    d8 = F.conv2d(ref_in[0], ref_in[3], ref_in[4]).view(n, a, 8, h, w).permute(0, 3, 4, 1, 2).reshape(n, -1, 8)
    g1, g2 = torch.randn(lo.shape, generator=gen), torch.randn(d8.shape, generator=gen)
    (lo * g1).sum().add((d8 * g2).sum()).backward()
    dev_in = [t.to(DEV).requires_grad_() for t in (x, wo, bo, wd, bd)]
    lo_d, d8_d = ops.rpn_head_1x1(*dev_in)
    assert lo_d.shape == lo.shape and d8_d.shape == d8.shape and lo_d.is_contiguous() and d8_d.is_contiguous()
    close(lo_d, lo, 1e-4, 1e-5, "flat logits")
    close(d8_d, d8, 1e-4, 1e-5, "flat deltas")
    (lo_d * g1.to(DEV)).sum().add((d8_d * g2.to(DEV)).sum()).backward()
    for nm, a_, b_ in zip(("dx", "dW_obj", "db_obj", "dW_delta", "db_delta"), dev_in, ref_in):
        close(a_.grad, b_.grad, 1e-4, 1e-4, "rpn head " + nm)
    with torch.no_grad():
        via = ops.conv1x1(dev_in[0], dev_in[1], dev_in[2]).permute(0, 2, 3, 1).reshape(n, -1)
    close(lo_d, via, 1e-5, 1e-6, "flat vs conv1x1 + permute")


def test_conv1x1(ops):
    gen = g(17)
    x = torch.randn(3, 96, 13, 21, generator=gen)
    w = torch.randn(72, 96, 1, 1, generator=gen) * 0.1
    b = torch.randn(72, generator=gen)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.conv2d(xr, wr, br)
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, w, b))
    yd = ops.conv1x1(xd, wd, bd)
    close(yd, yr, 1e-4, 1e-4, "1x1 fwd")
    yd.backward(gy.to(DEV))
    close(xd.grad, xr.grad, 1e-4, 1e-4, "1x1 dx")
    close(wd.grad, wr.grad, 1e-4, 1e-3, "1x1 dw")
    close(bd.grad, br.grad, 1e-4, 1e-3, "1x1 db")


# ------------------------------------------------------------------------------------------ ROIAlign
def _rand_boxes(gen, n, h, w, lo=4.0):
    cx, cy = torch.rand(n, generator=gen) * w, torch.rand(n, generator=gen) * h
    bw = lo + torch.rand(n, generator=gen) * w * 0.7
    bh = lo + torch.rand(n, generator=gen) * h * 0.7
    b = torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
    return b


@pytest.mark.parametrize("fh,fw", [(25, 31), (70, 27), (83, 83)])
def test_roi_align(ops, fh, fw):
    """(25, 31): the landscape layout of the grouped kernels; (70, 27) / (83, 83): maps with more than 64 rows (portrait and
    mixed-orientation batches) -- the grouped backward's four-piece table variant, fewer channel planes per workgroup."""
    gen = g(23)
    feat = torch.randn(2, 18, fh, fw, generator=gen)          # 18 channels: the grouped kernels' last group is ragged
    boxes = _rand_boxes(gen, 40, fh * 16, fw * 16)
    boxes[0] = torch.tensor([-30.0, -20.0, 40.0, 35.0])        # partly outside
    boxes[1] = torch.tensor([fw * 16 - 96.0, fh * 16 - 100.0, fw * 16 + 24.0, fh * 16 + 20.0])      # beyond the far border
    boxes[2] = torch.tensor([10.0, 10.0, 10.5, 10.2])          # tiny
    boxes[3] = torch.tensor([0.0, 0.0, fw * 16.0, fh * 16.0])  # whole image: many samples per bin (generic weight path)
    boxes[4] = torch.tensor([-9000.0, -7000.0, 9500.0, 8000.0])  # many times the map: a bin spans more cells than its table
    rois = torch.cat([torch.randint(0, 2, (40, 1), generator=gen).float(), boxes], 1)
    fr = feat.clone().requires_grad_()
    ref = d2.roi_align(fr, rois, 7, 1 / 16)
    gy = torch.randn(ref.shape, generator=gen)
    ref.backward(gy)
    fd = feat.to(DEV).requires_grad_()
    out = ops.roi_align(fd, rois.to(DEV), 7, 1 / 16)
    close(out, ref, 1e-5, 1e-5, "roi_align fwd")
    out.backward(gy.to(DEV))
    close(fd.grad, fr.grad, 1e-4, 1e-4, "roi_align bwd (ungrouped fallback: fp32 summation order differs from the CPU loop)")
    empty = ops.roi_align(fd, torch.zeros((0, 5), device=DEV), 7, 1 / 16)
    assert empty.shape == (0, 18, 7, 7)
    # grouped-by-image backward (LDS accumulation, no global atomics) gives the same gradient
    order = torch.argsort(rois[:, 0], stable=True)
    rs = rois[order]
    offs = torch.tensor([0, int((rs[:, 0] == 0).sum()), len(rs)], dtype=torch.int32, device=DEV)
    fd2 = feat.to(DEV).requires_grad_()
    out2 = ops.roi_align(fd2, rs.to(DEV), 7, 1 / 16, offs)
    # the grouped forward sums a bin's cells with separable weights (not the sample-by-sample order of the gather kernel)
    close(out2, out[order.to(DEV)], 1e-5, 1e-5, "grouped (LDS-plane) forward vs the gather kernel")
    close(out2, ref.detach()[order], 1e-5, 1e-5, "grouped forward vs the oracle")
    assert torch.equal(ops.roi_align(fd2, rs.to(DEV), 7, 1 / 16, offs), out2), "repeatable"
    out2.backward(gy[order].to(DEV))
    close(fd2.grad, fr.grad, 1e-4, 1e-4, "roi_align grouped bwd")
    fd3 = feat.to(DEV).requires_grad_()
    ops.roi_align(fd3, rs.to(DEV), 7, 1 / 16, offs).backward(gy[order].to(DEV))
    assert torch.equal(fd3.grad, fd2.grad), "grouped backward is atomic-free: must be bitwise reproducible"


# ------------------------------------------------------------------------------------------ boxes
def test_anchors_and_codec(ops):
    cell = d2.default_cell_anchors()
    ref = d2.grid_anchors(cell, (13, 21), 16, 0.0)
    got = ops.grid_anchors(cell.to(DEV), 13, 21, 16.0, 0.0)
    assert torch.equal(got.cpu(), ref), "anchors must be bit-exact"
    gen = g(31)
    src = _rand_boxes(gen, 500, 300, 400, lo=8.0)
    tgt = _rand_boxes(gen, 500, 300, 400, lo=8.0)
    for wts in ((1.0, 1.0, 1.0, 1.0), (10.0, 10.0, 5.0, 5.0)):
        ref_d = opt.get_deltas(src, tgt, wts)
        got_d = ops.get_deltas(src.to(DEV), tgt.to(DEV), wts)
        close(got_d, ref_d, 1e-5, 1e-5, "get_deltas")
        dl = torch.randn(500, 16, generator=gen)
        dl[0, 2] = 60.0
        ref_b = opt.apply_deltas(dl, src, wts)
        got_b = ops.apply_deltas(dl.to(DEV), src.to(DEV), wts, opt.SCALE_CLAMP)
        close(got_b, ref_b, 1e-5, 1e-3, "apply_deltas")
    # gradient of get_deltas w.r.t. the source boxes (differentiable anchors)
    s_r = src.clone().requires_grad_()
    opt.get_deltas(s_r, tgt, (1.0, 1.0, 1.0, 1.0)).mul(torch.arange(4.0) + 1).sum().backward()
    s_d = src.to(DEV).requires_grad_()
    ops.get_deltas(s_d, tgt.to(DEV), (1.0, 1.0, 1.0, 1.0)).mul(torch.arange(4.0, device=DEV) + 1).sum().backward()
    close(s_d.grad, s_r.grad, 1e-4, 1e-6, "get_deltas d/dsrc")


@pytest.mark.parametrize("m,nb,lowq,thr,labels", [(7, 5000, True, (0.3, 0.7), (0, -1, 1)),
                                                  (100, 37350, True, (0.3, 0.7), (0, -1, 1)),
                                                  (12, 2012, False, (0.5,), (0, 1)),
                                                  (0, 300, True, (0.3, 0.7), (0, -1, 1)),
                                                  (300, 700, False, (0.5,), (0, 1))])
def test_iou_match_bit_exact(ops, m, nb, lowq, thr, labels):
    gen = g(m + nb)
    if nb == 37350:
        boxes = d2.grid_anchors(d2.default_cell_anchors(), (50, 83), 16, 0.0)
        gt = _rand_boxes(gen, m, 800, 1333, lo=20.0).clamp_(min=0)
        gt[:, 2].clamp_(max=1333)
        gt[:, 3].clamp_(max=800)
    else:
        boxes = _rand_boxes(gen, nb, 300, 400, lo=5.0)
        gt = _rand_boxes(gen, m, 300, 400, lo=5.0) if m else torch.zeros((0, 4))
        if m:
            boxes[:m] = gt               # exact overlaps
            gt[-1] = torch.tensor([1000.0, 1000.0, 1010.0, 1010.0])   # a gt nobody overlaps (best IoU == 0)
    iou = d2.pairwise_iou(d2.Boxes(gt), d2.Boxes(boxes))
    ridx, rlab = d2.Matcher(list(thr), list(labels), allow_low_quality_matches=lowq)(iou)
    gidx, glab, giou = ops.iou_match(gt.to(DEV), boxes.to(DEV), thr, labels, lowq)
    assert torch.equal(glab.cpu(), rlab), f"labels differ at {(glab.cpu() != rlab).sum().item()} boxes"
    assert torch.equal(gidx.cpu(), ridx), "matched indices differ"
    if m:
        assert torch.equal(giou.cpu(), iou.max(dim=0)[0]), "IoU values must be bit-exact"


# ------------------------------------------------------------------------------------------ sort / nms / proposals
def test_segsort_desc(ops):
    gen = g(41)
    sizes = [37350, 16650, 0, 5, 12000]
    keys = torch.randn(sum(sizes), generator=gen)
    keys[10:20] = keys[10]                     # ties -> stable order
    offs = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32)
    sk, si = ops.segsort_desc(keys.to(DEV), offs.to(DEV))
    for a, b in zip(offs[:-1].tolist(), offs[1:].tolist()):
        rk, ri = torch.sort(keys[a:b], descending=True, stable=True)
        assert torch.equal(sk[a:b].cpu(), rk)
        assert torch.equal(si[a:b].cpu().long(), ri)


def _special_keys(n, gen):
    keys = torch.randn(n, generator=gen)
    if n >= 64:
        keys[10:20] = keys[10]                 # ties -> stable order
        keys[3], keys[n - 2] = float("inf"), float("inf")
        keys[5], keys[n // 2] = float("-inf"), float("-inf")
        keys[7], keys[n - 5] = 0.0, -0.0       # equal for a comparison sort: index order
        keys[n // 3] = float("nan")            # first in a descending torch.sort
        keys[40:60] = torch.floor(keys[40:60] * 2) / 2
    return keys


@pytest.mark.parametrize("sizes,topk", [([12000, 16384, 0, 5, 16000, 1, 127, 129, 128, 2048], None),
                                        ([37350, 20000, 16385, 300, 0, 16384], 12000), ([37350, 37350], 16384), ([65535], 6000)])
def test_segsort_lds_kernel(ops, sizes, topk):
    """the one-workgroup-per-segment LDS sort (bitonic network on (key, index) pairs; radix select + compaction for segments
    beyond 16 384 keys): whole segments / the first topk entries equal torch's stable descending sort, incl. ties, +-inf, +-0, NaN"""
    gen = g(sum(sizes) + (topk or 0))
    keys = torch.cat([_special_keys(n, gen) for n in sizes]) if sum(sizes) else torch.zeros(0)
    offs = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32)
    sk, si = ops.segsort_desc(keys.to(DEV), offs.to(DEV), max_len=max(sizes), topk=topk)
    sk, si = sk.cpu(), si.cpu().long()
    for a, b in zip(offs[:-1].tolist(), offs[1:].tolist()):
        rk, ri = torch.sort(keys[a:b], descending=True, stable=True)
        m = b - a if (b - a <= 16384 or topk is None) else topk
        assert torch.equal(si[a:a + m], ri[:m]), f"segment of {b - a}: order"
        assert torch.equal(sk[a:a + m].view(torch.int32), rk[:m].view(torch.int32)), f"segment of {b - a}: keys (bit patterns)"
        assert bool((sk[a + m:b] == float("-inf")).all()) and bool((si[a + m:b] == 0).all()), "entries past topk"
    assert torch.equal(ops.segsort_desc(keys.to(DEV), offs.to(DEV), max_len=max(sizes), topk=topk)[1].cpu().long(), si), "repeatable"


def test_efl_weight_of_near_uniform_teacher_rows_is_finite(ops):
    """Round 6: for near-uniform teacher rows (logit gap < ~1e-3) the base of the entropy-focal weight, 1 - H / log C, is >= 0 exactly
    but of the size of the device expf / logf rounding, and a base of -6e-8 to the power 0.5 is NaN -- it ended a loss-curve trajectory
    at iteration 452 (csrc/losses.hip::efl_weight clamps a negative base to 0 since).  Thousands of such rows, K = 1 and K = 8: both
    soft-label losses finite, gradients finite, and equal to the oracle's expression wherever THAT is finite (here: everywhere)."""
    gen = g(452)
    for C in (2, 9):
        r = 20000
        base = torch.randn(r, 1, generator=gen)
        T = base + torch.randn(r, C, generator=gen) * torch.logspace(-7, -2, r).unsqueeze(1)        # gaps 1e-7 .. 1e-2
        S = torch.randn(r, C, generator=gen)
        Sd = S.to(DEV).requires_grad_()
        got = ops.soft_ce_efl(T.to(DEV), Sd, 0.5, 0.5, True, 1.0 / r)
        got.backward()
        assert math.isfinite(float(got)) and bool(torch.isfinite(Sd.grad).all()), f"C = {C}: soft_ce_efl {float(got)}"
        p = F.softmax(T.double(), -1)
        ent = -(p * torch.log(p)).sum(-1)
        w = (1 - ent / math.log(C)).clamp_(min=0) ** 0.5
        ref = torch.sum(F.softmax(T.double() / 0.5, -1) * w.unsqueeze(-1) * -F.log_softmax(S.double(), -1)) / r
        # (the weights are square roots of a difference of two numbers that agree to 7 digits: in fp32 only their size survives)
        close(got, ref.float(), 0.3, 1e-6, f"C = {C}: soft ce on near-uniform rows")
        x = torch.randn(r, generator=gen).to(DEV).requires_grad_()
        l_cls, _ = ops.rpn_soft_obj_loss(T.to(DEV), x, 0.5, 0.5, True, 1.0 / r)
        l_cls.backward()
        assert math.isfinite(float(l_cls)) and bool(torch.isfinite(x.grad).all()), f"C = {C}: rpn_soft_obj_loss {float(l_cls)}"


def test_segsort_host_lengths_guard(ops):
    """a max_len / topk that does not describe the segments is a PtmiError BEFORE the launch when the caller hands over
    its host-side segment lengths (all three product call sites do) -- not NaN keys that only a training-mode check would notice"""
    from probabilisticteacher_amd._lib import PtmiError
    sizes = [300, 17000, 5]
    keys = torch.randn(sum(sizes), generator=g(9))
    offs = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32)
    with pytest.raises(PtmiError, match="segment lengths"):
        ops.segsort_desc(keys.to(DEV), offs.to(DEV), max_len=16384, lengths=sizes)            # longest segment > max_len
    with pytest.raises(PtmiError, match="segment lengths"):
        ops.segsort_desc(keys.to(DEV), offs.to(DEV), max_len=17000, lengths=[300, 17000])     # wrong segment count
    sk, si = ops.segsort_desc(keys.to(DEV), offs.to(DEV), lengths=sizes)                       # max_len derived from the lengths
    for a, b in zip(offs[:-1].tolist(), offs[1:].tolist()):
        rk, ri = torch.sort(keys[a:b], descending=True, stable=True)
        assert torch.equal(si.cpu().long()[a:b], ri) and torch.equal(sk.cpu()[a:b], rk)


@pytest.mark.parametrize("counts,thr,max_keep", [([300, 0, 1, 65, 1000], 0.7, 2000), ([12000, 9000], 0.7, 2000),
                                                 ([5000], 0.5, 100)])
def test_nms_bit_exact(ops, counts, thr, max_keep):
    gen = g(sum(counts))
    boxes_all, keep_ref = [], []
    for c in counts:
        ctr = torch.rand(c, 2, generator=gen) * torch.tensor([1333.0, 800.0])
        wh = torch.exp(torch.rand(c, 2, generator=gen) * 3.0 + 2.5)
        b = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
        sc = torch.rand(c, generator=gen)
        order = d2.descending_order(sc)
        b = b[order]                                    # the HIP API takes boxes already in descending-score order
        boxes_all.append(b)
        keep_ref.append(d2.nms(b, torch.arange(c, 0, -1).float(), thr)[:max_keep])
    allb = torch.cat(boxes_all, 0) if sum(counts) else torch.zeros((0, 4))
    offs = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32)
    keep, cnt = ops.nms_batched(allb.to(DEV), offs.to(DEV), max(counts), thr, max_keep)
    for i, ref in enumerate(keep_ref):
        k = int(cnt[i])
        assert k == len(ref), f"image {i}: kept {k} vs {len(ref)}"
        assert torch.equal(keep[i, :k].cpu().long(), ref), f"image {i}: keep lists differ"


@pytest.mark.parametrize("thr", [0.5, 0.7, 0.3])
def test_nms_iou_at_the_threshold(ops, thr):
    """Pairs of boxes whose IoU equals the threshold exactly (integer boxes: inter / union = thr) or misses it by a few
    ulps of one coordinate: the kernel decides most pairs from `inter` vs `thr * union` and divides only when that cannot
    decide -- the suppression decisions must stay those of the correctly rounded fp32 quotient (strict `>`), bit for bit."""
    rs = np.random.RandomState(int(thr * 100))
    num, den = {0.5: (1, 2), 0.7: (7, 10), 0.3: (3, 10)}[thr]
    boxes = []
    for k in range(400):
        # A = [x, y, x + den * s, y + 1 * t]; B overlaps A on `num` of its `den` columns and has the same size:
        # inter = num s t, union = (2 den - num) s t -> generalise: choose B so that inter / union = num / den
        s_, t_ = float(rs.randint(1, 40)), float(rs.randint(1, 30))
        x, y = float(rs.randint(0, 500) + 1000 * (k % 20)), float(rs.randint(0, 300) + 1000 * (k // 20))
        # A has width den*s, B = A's sub-box of width num*s (inside A): inter = area(B), union = area(A) -> IoU = num/den
        a = np.array([x, y, x + den * s_, y + t_], np.float32)
        b = np.array([x, y, x + num * s_, y + t_], np.float32)
        j = rs.randint(0, 5)
        if j:                                         # move one coordinate of B by -2 .. +2 ulps
            c = rs.randint(0, 4)
            for _ in range(abs(j - 2) if j != 2 else 0):
                b[c] = np.nextafter(b[c], np.float32(np.inf if j > 2 else -np.inf))
        boxes += [a, b]
    b = torch.from_numpy(np.stack(boxes))
    n = b.shape[0]
    ref = d2.nms(b, torch.arange(n, 0, -1).float(), thr)
    keep, cnt = ops.nms_batched(b.to(DEV), torch.tensor([0, n], dtype=torch.int32, device=DEV), n, thr, n)
    k = int(cnt[0])
    assert k == len(ref) and torch.equal(keep[0, :k].cpu().long(), ref), f"kept {k} vs {len(ref)}"
    assert 0 < n - k < n // 2 + 1                     # some pairs suppress, some do not


def test_nms_no_candidates_at_all(ops):
    """every image of the batch has zero candidates (a teacher without any detection above the score threshold)"""
    seg = torch.zeros(4, dtype=torch.int32, device=DEV)
    keep, cnt = ops.nms_batched(torch.zeros((0, 4), device=DEV), seg, 0, 0.5, 100)
    assert keep.shape == (3, 100) and cnt.cpu().tolist() == [0, 0, 0]


def test_nms_matches_bruteforce_small(ops):
    gen = g(3)
    b = _rand_boxes(gen, 200, 100, 100, lo=5.0)
    sc = torch.rand(200, generator=gen)
    order = d2.descending_order(sc)
    ref = d2.nms_bruteforce(b.numpy(), sc.numpy(), 0.5)
    keep, cnt = ops.nms_batched(b[order].to(DEV), torch.tensor([0, 200], dtype=torch.int32, device=DEV), 200, 0.5, 200)
    got = order[keep[0, :int(cnt[0])].cpu().long()]
    assert np.array_equal(got.numpy(), ref)


def test_rpn_prepare(ops):
    gen = g(53)
    n, r, k = 2, 3000, 1200
    dec = torch.randn(n, r, 4, generator=gen) * 200 + 300
    dec[..., 2:] = dec[..., :2] + torch.rand(n, r, 2, generator=gen) * 300 - 20     # some empty boxes
    logits = torch.randn(n, r, generator=gen)
    sigma = torch.randn(n, r, 4, generator=gen)
    sizes = torch.tensor([[600.0, 800.0], [544.0, 720.0]])
    srt, idx = logits.sort(descending=True, dim=1, stable=True)
    boxes, keys, counts, nonfin = ops.rpn_prepare(dec.to(DEV), srt.to(DEV), idx.int().to(DEV), sigma.to(DEV),
                                                  sizes.to(DEV), k, 0.0)
    for i in range(n):
        bx = d2.Boxes(dec[i][idx[i, :k]].clone())
        bx.clip((int(sizes[i, 0]), int(sizes[i, 1])))
        assert torch.equal(boxes[i].cpu(), bx.tensor), "clipped boxes must be bit-exact"
        ne = bx.nonempty(0.0)
        valid = keys[i].cpu() != float("-inf")
        assert torch.equal(valid, ne), "nonempty mask must be bit-exact"
        assert int(counts[i]) == int(ne.sum()) and 0 < int(ne.sum()) < k, "kept count"
        ref_sc = srt[i, :k] * (1 - torch.sigmoid(sigma[i, :k]).sum(-1) / 4.0)
        close(keys[i].cpu()[valid], ref_sc[valid], 1e-5, 1e-6, "rescoring")
    assert int(nonfin.sum()) == 0
    dec[1, idx[1, 3], 0] = float("inf")
    _, keys2, counts2, nonfin2 = ops.rpn_prepare(dec.to(DEV), srt.to(DEV), idx.int().to(DEV), sigma.to(DEV),
                                                 sizes.to(DEV), k, 0.0)
    assert nonfin2.cpu().tolist() == [0, 1] and float(keys2[1, 3]) == float("-inf")
    assert int(counts2[1]) == int(counts[1]) - int(keys[1, 3] != float("-inf"))
    # k < 64: a wave spans several images (per-image counting peels one image at a time)
    n, r, k = 5, 40, 17
    dec = torch.rand(n, r, 4, generator=gen) * 50
    dec[..., 2:] = dec[..., :2] + torch.rand(n, r, 2, generator=gen) * 40 - 10
    srt, idx = torch.randn(n, r, generator=gen).sort(descending=True, dim=1, stable=True)
    sizes = torch.tensor([[64.0, 64.0]] * n)
    boxes, keys, counts, _ = ops.rpn_prepare(dec.to(DEV), srt.to(DEV), idx.int().to(DEV), torch.zeros(n, r, 4, device=DEV),
                                             sizes.to(DEV), k, 0.0)
    assert counts.cpu().tolist() == (keys.cpu() != float("-inf")).sum(1).tolist()


def test_nms_with_device_side_counts(ops):
    """fixed-capacity segments: only the first seg_counts[i] boxes of a segment exist"""
    gen = g(77)
    cap, fills, thr = 700, [700, 0, 333, 64, 1], 0.6
    allb, refs = [], []
    for f in fills:
        b = _rand_boxes(gen, cap, 300, 300, lo=8.0)
        allb.append(b)
        refs.append(d2.nms(b[:f], torch.arange(f, 0, -1).float(), thr)[:50] if f else torch.zeros(0, dtype=torch.int64))
    seg = torch.arange(0, (len(fills) + 1) * cap, cap, dtype=torch.int32, device=DEV)
    keep, cnt = ops.nms_batched(torch.cat(allb).to(DEV), seg, cap, thr, 50,
                                seg_counts=torch.tensor(fills, dtype=torch.int32, device=DEV))
    for i, ref in enumerate(refs):
        assert int(cnt[i]) == len(ref) and torch.equal(keep[i, :len(ref)].cpu().long(), ref), f"segment {i}"


# ------------------------------------------------------------------------------------------ losses
def test_bce_and_gaussian_nll(ops):
    gen = g(61)
    x = torch.randn(5000, generator=gen) * 3
    lab = torch.randint(-1, 2, (5000,), generator=gen).to(torch.int8)
    xr = x.clone().requires_grad_()
    valid = lab >= 0
    ref = F.binary_cross_entropy_with_logits(xr[valid], lab[valid].float(), reduction="sum") / 512.0
    ref.backward()
    xd = x.to(DEV).requires_grad_()
    got = ops.bce_logits_sum(xd, lab.to(DEV), 1 / 512.0)
    close(got, ref, 1e-5, 1e-6, "bce")
    (got * 1.0).backward()
    close(xd.grad, xr.grad, 1e-4, 1e-7, "bce grad")

    d = torch.randn(700, 8, generator=gen)
    t = torch.randn(700, 4, generator=gen)
    dr, tr = d.clone().requires_grad_(), t.clone().requires_grad_()
    ref = -torch.log(opt.gaussian_dist_pdf(dr[:, :4], tr, torch.sigmoid(dr[:, 4:])) + 1e-9).sum() / 256.0
    ref.backward()
    dd, td = d.to(DEV).requires_grad_(), t.to(DEV).requires_grad_()
    got = ops.gaussian_nll_sum(dd, td, 1 / 256.0)
    close(got, ref, 2e-5, 1e-6, "gaussian nll")
    (got * 2.0).backward()
    close(dd.grad, 2 * dr.grad, 2e-4, 1e-6, "gaussian nll dd")
    close(td.grad, 2 * tr.grad, 2e-4, 1e-6, "gaussian nll dt")
    zero = ops.gaussian_nll_sum(torch.zeros((0, 8), device=DEV), torch.zeros((0, 4), device=DEV), 1.0)
    assert float(zero) == 0.0


def test_softmax_ce(ops):
    gen = g(67)
    x = torch.randn(513, 9, generator=gen) * 2
    t = torch.randint(0, 9, (513,), generator=gen)
    xr = x.clone().requires_grad_()
    ref = F.cross_entropy(xr, t)
    ref.backward()
    xd = x.to(DEV).requires_grad_()
    got = ops.softmax_ce_mean(xd, t.to(DEV))
    close(got, ref, 1e-5, 1e-6, "ce")
    got.backward()
    close(xd.grad, xr.grad, 1e-4, 1e-8, "ce grad")
    close(ops.softmax_rows(x.to(DEV)), F.softmax(x, -1), 1e-5, 1e-7, "softmax")
    assert float(ops.softmax_ce_mean(torch.zeros((0, 9), device=DEV), torch.zeros(0, dtype=torch.int64, device=DEV))) == 0


def test_unsup_losses(ops):
    """The three entropy-focal soft-label losses against the oracle restatement (which is pinned to the
    reference by tests/golden)."""
    gen = g(71)
    cfg = opt.Cfg(tau=(0.5, 0.25), efl_lambda=(0.5, 0.5))
    K = 8
    # ROI soft CE
    T = torch.randn(300, K + 1, generator=gen) * 2
    S = torch.randn(300, K + 1, generator=gen)
    Sr = S.clone().requires_grad_()
    nls = -F.log_softmax(Sr, -1)
    p = F.softmax(T, -1)
    ent = -(p * torch.log(p)).sum(-1)
    q = F.softmax(T / cfg.tau[0], -1) * ((1 - ent / math.log(K + 1)) ** cfg.efl_lambda[0]).unsqueeze(-1)
    ref = torch.sum(q * nls) / 300
    ref.backward()
    Sd = S.to(DEV).requires_grad_()
    got = ops.soft_ce_efl(T.to(DEV), Sd, cfg.tau[0], cfg.efl_lambda[0], True, 1 / 300)
    close(got, ref, 2e-5, 1e-6, "soft ce")
    got.backward()
    close(Sd.grad, Sr.grad, 2e-4, 1e-8, "soft ce grad")

    # RPN soft objectness + KL, via the oracle's rpn_losses_unsup on a synthetic single image
    A = 600
    anchors = _rand_boxes(gen, A, 300, 400, lo=16.0)
    obj = torch.randn(1, A, generator=gen)
    deltas = torch.randn(1, A, 8, generator=gen)
    mask = torch.rand(A, generator=gen) < 0.3
    k = int(mask.sum())
    soft = torch.randn(k, K + 1, generator=gen) * 2
    sig = torch.randn(k, 4, generator=gen)
    matched = _rand_boxes(gen, A, 300, 400, lo=16.0)
    objr, dr = obj.clone().requires_grad_(), deltas.clone().requires_grad_()
    ref = opt.rpn_losses_unsup(cfg, anchors, objr, dr, [soft], [mask], [matched], [sig], True)
    (ref["loss_rpn_cls"] + ref["loss_rpn_loc"]).backward()
    inv = 1.0 / (cfg.rpn_batch_size_per_image * 1)
    xd = obj[0][mask].to(DEV).requires_grad_()
    l_cls, fg = ops.rpn_soft_obj_loss(soft.to(DEV), xd, cfg.tau[0], cfg.efl_lambda[0], True, inv)
    close(l_cls, ref["loss_rpn_cls"], 2e-5, 1e-6, "rpn soft obj")
    fg_ref = soft.max(-1)[1] != K
    assert torch.equal(fg.cpu().bool(), fg_ref)
    qd = deltas[0][mask].to(DEV).requires_grad_()
    mu_p = opt.get_deltas(anchors, matched, cfg.rpn_bbox_reg_weights)[mask]
    l_loc = ops.kl_efl_loss(qd, mu_p.to(DEV), sig.to(DEV), fg, cfg.tau[1], cfg.efl_lambda[1], True, 0, inv)
    close(l_loc, ref["loss_rpn_loc"], 5e-5, 1e-6, "rpn kl")
    (l_cls + l_loc).backward()
    close(xd.grad, objr.grad[0][mask], 2e-4, 1e-8, "rpn soft obj grad")
    close(qd.grad, dr.grad[0][mask], 5e-4, 1e-7, "rpn kl grad")
    # ROI flavour: mean reduction over the selected rows
    mu_q = deltas[0][mask][:, :4]
    var_p = torch.sigmoid(sig)
    entb = 0.5 * torch.log(2 * np.pi * np.e * var_p)
    wb = (1 - entb / (0.5 * math.log(2 * np.pi * np.e))) ** cfg.efl_lambda[1]
    var_q = torch.sigmoid(deltas[0][mask][:, 4:])
    vp = var_p * cfg.tau[1]
    kl = (0.5 * torch.log(var_q / vp) - 0.5 + (vp + (mu_q - mu_p) ** 2) / (2 * var_q)) * wb
    l_mean = ops.kl_efl_loss(deltas[0][mask].to(DEV), mu_p.to(DEV), sig.to(DEV), None, cfg.tau[1], cfg.efl_lambda[1],
                             True, 1, 1.0)
    close(l_mean, kl.mean(), 5e-5, 1e-6, "roi kl mean")


# ------------------------------------------------------------------------------------------ optimiser / image prep
def test_ema_clip_sgd(ops):
    gen = g(83)
    n = 100003
    s, t = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
    td = t.to(DEV)
    ops.ema_update(s.to(DEV), td, 0.9996)
    ref = opt.ema_update({"p": s}, {"p": t.clone()}, 0.9996)["p"]
    assert torch.equal(td.cpu(), ref), "EMA must be bit-exact (same op order, no FMA contraction)"
    gr = torch.randn(n, generator=gen) * 0.2
    ss = ops.sumsq(gr.to(DEV))
    close(ss.reshape(()), (gr.double() ** 2).sum().float(), 1e-5, 0, "sumsq")
    p, buf = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
    for first in (True, False):
        pd, bd = p.to(DEV), buf.to(DEV)
        ops.clip_sgd_step(pd, gr.to(DEV), bd, ss, 10.0, 0.016, 0.9, 1e-4, first)
        sc = 10.0 / max(float(gr.norm()), 10.0)
        g2 = gr * sc + 1e-4 * p
        b2 = g2 if first else 0.9 * buf + g2
        close(bd, b2, 1e-5, 1e-6, "momentum")
        close(pd, p - 0.016 * b2, 1e-5, 1e-6, "param")


def _paste_ref(cfg, img, ratio):
    """canvas of trainer.py:563-576 with the shrink computed by the oracle's C restatement of ATen's evaluation order"""
    h, w = img.shape[-2:]
    dh, dw = int(h * ratio), int(w * ratio)
    x1, y1 = int((w - dw) / 2), int((h - dh) / 2)
    bg = torch.zeros_like(img)
    bg += torch.tensor(cfg.pixel_mean).view(3, 1, 1).int()
    bg[:, y1:y1 + dh, x1:x1 + dw] = d2.bilinear_shrink_u8(img, dh, dw)
    return bg, x1, y1


def test_preprocess_and_shrink_paste(ops):
    """Byte outputs are BIT-EXACT: preprocess vs the oracle; shrink_paste vs (a) the uint8 canvases the REAL reference
    produced (tests/golden/trainer_pieces.npz), (b) the oracle at toy and BASELINE (1333x800) size, single-image and
    whole-batch entry points."""
    cfg = opt.Cfg()
    mean_int = [int(m) for m in cfg.pixel_mean]
    rs = np.random.RandomState(1)
    imgs = [torch.from_numpy(rs.randint(0, 256, (3, 50, 70)).astype(np.uint8)),
            torch.from_numpy(rs.randint(0, 256, (3, 44, 61)).astype(np.uint8))]
    ref = opt.preprocess_image(cfg, [{"image": im} for im in imgs]).tensor
    got = ops.preprocess_images([im.to(DEV) for im in imgs], cfg.pixel_mean, cfg.pixel_std)      # batched launch
    assert torch.equal(got.cpu(), ref), "preprocess must be bit-exact"
    one = torch.empty((3, 50, 70), device=DEV)
    im0 = imgs[0].to(DEV)
    call, _p, _stream = _raw()
    call("ptmi_preprocess_image", _p(im0), _p(one), 50, 70, 50, 70, *[float(v) for v in cfg.pixel_mean],
         *[float(v) for v in cfg.pixel_std], _stream())
    assert torch.equal(one.cpu(), ref[0]), "single-image preprocess must be bit-exact"
    # (a) the real reference's PTrainer.resize outputs
    z = load("trainer_pieces")
    recs = records(z, "rz_in", 2)
    for i, (r, q) in enumerate(zip(recs, z["rz_ratios"])):
        out, _, _ = ops.shrink_paste(r["image"].to(DEV), float(q), mean_int)
        assert np.array_equal(out.cpu().numpy(), z[f"rz_out{i}_image"]), "shrink_paste vs the reference's canvas"
    outs, _ = ops.shrink_paste_batch([r["image"].to(DEV) for r in recs], [float(q) for q in z["rz_ratios"]], mean_int)
    for i, o in enumerate(outs):
        assert np.array_equal(o.cpu().numpy(), z[f"rz_out{i}_image"]), "batched shrink_paste vs the reference's canvas"
    # (b) oracle, toy sizes (incl. ratio 1.0: identity copy) and BASELINE size
    big = torch.from_numpy(rs.randint(0, 256, (3, 800, 1333)).astype(np.uint8))
    cases = [(imgs[0], r) for r in (0.731, 0.5, 0.999, 1.0)] + [(imgs[1], 0.6180339)] + [(big, r) for r in (0.5, 0.83, 0.99999)]
    for im, ratio in cases:
        want, x1r, y1r = _paste_ref(cfg, im, ratio)
        out, x1, y1 = ops.shrink_paste(im.to(DEV), ratio, mean_int)
        assert (x1, y1) == (x1r, y1r) and torch.equal(out.cpu(), want), f"shrink_paste {tuple(im.shape)} ratio {ratio}"
    outs, offs = ops.shrink_paste_batch([im.to(DEV) for im, _ in cases], [r for _, r in cases], mean_int)
    for (im, ratio), o in zip(cases, outs):
        assert torch.equal(o.cpu(), _paste_ref(cfg, im, ratio)[0]), f"batched shrink_paste ratio {ratio}"


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 37, 45), (1, 128, 128, 20, 83), (1, 3, 64, 24, 33),
                                            (1, 16, 64, 11, 666), (1, 8, 130, 6, 601)])      # wide: row-major blocks
def test_conv3x3_fused_relu_pool(ops, n, cin, cout, h, w):
    gen = g(cin + h)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    ref = F.max_pool2d(F.relu(F.conv2d(x, wt, b, padding=1)), 2, 2)
    got = ops.conv3x3_relu_pool_nograd(x.to(DEV), wt.to(DEV), b.to(DEV))
    close(got, ref, 1e-4, 1e-4, "fused conv+relu+pool")
    with torch.no_grad():
        unfused = ops.maxpool2x2(ops.conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True))
    if cin > 4:
        assert torch.equal(got, unfused), "fused epilogue must equal conv -> pool bit for bit"
    else:       # the unfused 3-channel conv runs on the VALU stem kernel (different summation order than the MFMA path)
        close(got, unfused, 1e-5, 1e-5, "fused vs unfused stem")


@pytest.mark.parametrize("pool", [True, False])
def test_vgg_block_fused_backward(ops, pool):
    """The single-node VGG block (fused ReLU masks in dgrad / pool backward) vs stock torch autograd."""
    gen = g(97)
    x = torch.randn(2, 16, 22, 37, generator=gen)
    ws = [torch.randn(24, 16, 3, 3, generator=gen) * 0.1, torch.randn(24, 24, 3, 3, generator=gen) * 0.08,
          torch.randn(32, 24, 3, 3, generator=gen) * 0.08]
    bs = [torch.randn(c, generator=gen) * 0.1 for c in (24, 24, 32)]
    xr = x.clone().requires_grad_()
    wr = [t.clone().requires_grad_() for t in ws]
    br = [t.clone().requires_grad_() for t in bs]
    y = xr
    for w_, b_ in zip(wr, br):
        y = F.relu(F.conv2d(y, w_, b_, padding=1))
    if pool:
        y = F.max_pool2d(y, 2, 2)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)
    xd = x.to(DEV).requires_grad_()
    wd = [t.to(DEV).requires_grad_() for t in ws]
    bd = [t.to(DEV).requires_grad_() for t in bs]
    params = [t for pair in zip(wd, bd) for t in pair]
    yd = ops.vgg_block(xd, pool, params)
    close(yd, y, 1e-4, 1e-4, "block fwd")
    yd.backward(gy.to(DEV))
    close(xd.grad, xr.grad, 1e-4, 2e-4, "block dx")
    for i in range(3):
        close(wd[i].grad, wr[i].grad, 1e-4, 1e-3, f"block dw{i}")
        close(bd[i].grad, br[i].grad, 1e-4, 1e-3, f"block db{i}")


# ------------------------------------------------------------------------------------------ shape fuzz
def test_conv3x3_random_shapes(ops):
    """24 seeded random shapes (odd widths and heights down to 1, channel counts off the tile sizes, batch 1-3)
    through forward, dgrad, wgrad and bias gradient against torch CPU: edge-tile, halo and range-check handling."""
    rng = np.random.RandomState(2024)
    for case in range(24):
        n = int(rng.randint(1, 4))
        cin = int(rng.choice([1, 3, 4, 5, 17, 32, 33, 64]))
        cout = int(rng.choice([1, 7, 32, 64, 65, 128, 130]))
        h = int(rng.choice([1, 2, 3, 4, 5, 9, 13]))
        w = int(rng.choice([1, 2, 3, 5, 8, 25, 31, 32, 33, 39, 57, 64, 71]))
        relu = bool(rng.randint(0, 2))
        gen = g(1000 + case)
        x = torch.randn(n, cin, h, w, generator=gen)
        wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
        b = torch.randn(cout, generator=gen) * 0.1
        xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), b.clone().requires_grad_()
        yr = F.conv2d(xr, wr, br, padding=1)
        if relu:
            yr = F.relu(yr)
        gy = torch.randn(yr.shape, generator=gen)
        yr.backward(gy)
        xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, wt, b))
        yd = ops.conv3x3(xd, wd, bd, relu)
        tag = f"case {case} n={n} cin={cin} cout={cout} h={h} w={w} relu={relu}"
        close(yd, yr, 1e-4, 1e-4, "fwd " + tag)
        yd.backward(gy.to(DEV))
        close(xd.grad, xr.grad, 1e-4, 2e-4, "dgrad " + tag)
        close(wd.grad, wr.grad, 1e-4, 1e-3, "wgrad " + tag)
        close(bd.grad, br.grad, 1e-4, 1e-3, "bias grad " + tag)


@pytest.mark.parametrize("n,cin,cout,h,w", [
    (2, 3, 64, 5, 1),       # one-column image: a lane's four flat pixels are four rows (four frames)
    (1, 3, 64, 7, 2),       # two frames per lane
    (2, 3, 64, 9, 3),       # lanes straddle one or two row ends
    (1, 3, 64, 6, 5),       # H W % 4 != 0: ragged last lane, unaligned planes
    (3, 3, 64, 11, 67),     # H W = 737: planes at every 4-B misalignment, row ends inside many lanes
    (1, 3, 64, 3, 257),     # a row longer than one 256-pixel block
    (2, 1, 20, 4, 9),       # Cin = 1, channel count off the 16-channel wave groups
    (1, 2, 70, 8, 13),      # Cin = 2, two 64-channel tiles, the second ragged
    (2, 4, 130, 5, 31),     # Cin = 4 (the other template instance), three tiles
    (1, 3, 64, 1, 1),       # a single pixel
])
@pytest.mark.parametrize("relu", [False, True])
def test_conv3x3_stem_flat_mapping(ops, n, cin, cout, h, w, relu):
    """The Cin <= 4 stem kernel tiles the FLATTENED plane (four consecutive flat pixels per lane, re-run per row
    touched): forward against torch CPU on shapes whose lanes wrap over row ends, whose planes are not 16-B aligned,
    and whose channel counts do not fill the wave's channel groups.  Also checks nothing is written past the tensor."""
    gen = g(31 * h + w + cin)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    yr = F.conv2d(x, wt, b, padding=1)
    if relu:
        yr = F.relu(yr)
    wp = ops.conv3x3_pack(wt.to(DEV), 0, 1 if relu else 0)
    guard = torch.full((n * cout * h * w + 64,), 7.5, device=DEV)
    yd = guard[:n * cout * h * w].view(n, cout, h, w)
    from probabilisticteacher_amd import _lib
    xd, bd = x.to(DEV), b.to(DEV)
    _lib.call("ptmi_conv3x3_fwd", ops._ptr(xd), ops._ptr(wp), ops._ptr(bd), None, ops._ptr(yd), n, cin, cout, h, w,
              1 if relu else 0, ops._stream())
    torch.cuda.synchronize()
    close(yd, yr, 1e-5, 1e-5, "stem fwd")
    assert (guard[n * cout * h * w:] == 7.5).all(), "stem wrote past the end of the output"


# ------------------------------------------------------------------------------------------ batched matching / sampling
def test_iou_match_batched_equals_per_image(ops):
    """ptmi_iou_match_batched (one launch pair per batch) against ptmi_iou_match image by image: indices, labels and
    IoU values bit for bit, with per-image box sets, a shared box set (RPN anchors) and images without ground truth."""
    gen = g(77)
    n = 4
    gts = [_rand_boxes(gen, m, 300, 400) for m in (5, 0, 1, 17)]
    props = [_rand_boxes(gen, p, 300, 400) for p in (700, 33, 0, 1200)]
    gt_all = torch.cat(gts).to(DEV)
    boxes_all = torch.cat(props).to(DEV)
    for lowq, thr, labs in ((False, (0.5,), (0, 1)), (True, (0.3, 0.7), (0, -1, 1))):
        midx, mlab, miou, _, _ = ops.iou_match_batched(gt_all, [len(x) for x in gts], boxes_all, [len(x) for x in props],
                                                       thr, labs, lowq)
        o = 0
        for gt, pb in zip(gts, props):
            if len(pb):
                ri, rl, ru = ops.iou_match(gt.to(DEV), pb.to(DEV), thr, labs, lowq)
                assert torch.equal(midx[o:o + len(pb)], ri) and torch.equal(mlab[o:o + len(pb)], rl)
                assert torch.equal(miou[o:o + len(pb)], ru)
            o += len(pb)
        anchors = _rand_boxes(gen, 3000, 300, 400).to(DEV)
        midx, mlab, miou, _, _ = ops.iou_match_batched(gt_all, [len(x) for x in gts], anchors, None, thr, labs, lowq)
        assert midx.shape == (n, 3000)
        for i, gt in enumerate(gts):
            ri, rl, ru = ops.iou_match(gt.to(DEV), anchors, thr, labs, lowq)
            assert torch.equal(midx[i], ri) and torch.equal(mlab[i], rl) and torch.equal(miou[i], ru)


def test_sample_by_keys_equals_reference_subsample_labels(ops):
    """ptmi_sample_by_keys against D2's subsample_labels driven by the same keys (oracle KeyedPerm): sample sizes, members
    and order, for images with many / few / no foreground candidates and fewer background candidates than requested."""
    gen = g(78)
    K = 8
    sizes, n_fgs = (2007, 640, 100, 1), (300, 12, 0, 1)
    cls_list = []
    for p_count, n_fg in zip(sizes, n_fgs):
        cls = torch.full((p_count,), K, dtype=torch.int64)
        cls[torch.randperm(p_count, generator=gen)[:n_fg]] = torch.randint(0, K, (n_fg,), generator=gen)
        cls[torch.randperm(p_count, generator=gen)[:p_count // 10]] = -1
        cls_list.append(cls)
    kp = opt.KeyedPerm(9)
    keys = torch.cat([kp.draw((s,)) for s in sizes])
    keys[5] = keys[9]                                        # a tie: broken by index, like a stable argsort
    kp.pending[0][5] = kp.pending[0][9]
    off = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device=DEV)
    fg, bg, cnt = ops.sample_by_keys(torch.cat(cls_list).to(DEV), keys.to(DEV), off, max(sizes), 512, 128, K)
    cnt = cnt.cpu().tolist()
    kp.start_replay()
    for i, cls in enumerate(cls_list):
        rf, rb = d2.subsample_labels(cls, 512, 0.25, K, kp)
        assert cnt[i] == [len(rf), len(rb)], (i, cnt[i], len(rf), len(rb))
        assert torch.equal(fg[i, :cnt[i][0]].cpu(), rf) and torch.equal(bg[i, :cnt[i][1]].cpu(), rb), i


@pytest.mark.parametrize("m,n,k,ta,tb,bias_mode,relu", [
    (70, 50, 300, 0, 1, 2, True),          # Linear forward (x (R,K), W (N,K))
    (130, 257, 96, 0, 0, 0, False),        # Linear dX
    (200, 131, 515, 1, 0, 0, False),       # Linear dW (both operands mn-fast), K % 16 != 0
    (64, 300, 45, 1, 1, 1, False),         # the fourth layout, K < 64 with a ragged tail
    (1024, 256, 4096, 1, 0, 0, False),     # long K, few tiles: the split-K path
    (33, 20, 7, 0, 1, 2, False),           # K smaller than one MFMA step
])
def test_gemm_bf16_native(ops, m, n, k, ta, tb, bias_mode, relu):
    """ptmi_gemm_bf16 (v_mfma_f32_32x32x16_bf16, operands rounded between LDS and the MFMA) against the fp32 product of
    the bf16-rounded operands computed in float64: every layout, ragged M / N / K, bias / ReLU epilogues, split-K."""
    gen = g(m + n + k)
    A = torch.randn((k, m) if ta else (m, k), generator=gen)
    B = torch.randn((n, k) if tb else (k, n), generator=gen)
    bias = torch.randn(m if bias_mode == 1 else n, generator=gen) if bias_mode else None
    rb = lambda t: t.to(torch.bfloat16).double()
    ref = (rb(A).t() if ta else rb(A)) @ (rb(B).t() if tb else rb(B))
    if bias_mode == 1:
        ref = ref + bias.double()[:, None]
    elif bias_mode == 2:
        ref = ref + bias.double()[None, :]
    if relu:
        ref = ref.clamp_min(0)
    ops.set_operand_rounding("bf16")
    try:
        out = ops.gemm(A.to(DEV), B.to(DEV), m, n, k, A.shape[1], B.shape[1], ta, tb,
                       bias=None if bias is None else bias.to(DEV), bias_mode=bias_mode, relu=relu)
        out2 = ops.gemm(A.to(DEV), B.to(DEV), m, n, k, A.shape[1], B.shape[1], ta, tb,
                        bias=None if bias is None else bias.to(DEV), bias_mode=bias_mode, relu=relu)
    finally:
        ops.set_operand_rounding(None)
    close(out, ref.float(), 1e-5, 4e-6 * math.sqrt(k) + 1e-5, "bf16 gemm")      # fp32 accumulation of k unit-scale products
    assert torch.equal(out, out2), "bf16 gemm is not reproducible run to run"
    f32 = ops.gemm(A.to(DEV), B.to(DEV), m, n, k, A.shape[1], B.shape[1], ta, tb,
                   bias=None if bias is None else bias.to(DEV), bias_mode=bias_mode, relu=relu)
    if k >= 64:
        assert float((f32 - out).abs().max()) > 1e-3, "the bf16 entry point did not round its operands"


def _rb(t):
    return t.to(torch.bfloat16).float()


def close_stored_bf16(a, b, what):
    """a = a tensor the bf16-STORAGE kernels produced (csrc/p8.hip: the fp32 accumulator rounded to bf16 when it is stored),
    b = the fp32 reference of the same quantity: one bf16 ulp (2^-8 relative; a rounding boundary may be crossed) on top of the
    fp32 summation-order bar"""
    a, b = a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy()
    assert a.shape == b.shape, f"{what}: {a.shape} vs {b.shape}"
    err = np.abs(a - b)
    tol = 2.0 ** -7 * np.abs(b) + 2e-4 * max(float(np.abs(b).max()), 1e-30)
    assert (err <= tol).all(), f"{what}: max abs err {err.max():.3e}, max |ref| {np.abs(b).max():.3e}, {int((err > tol).sum())} of {err.size} off"


@pytest.mark.parametrize("n,cin,cout,h,w,relu", [
    (2, 3, 64, 37, 45, True),        # the stem layer: 3 channels padded to one 16-channel chunk
    (1, 64, 64, 24, 40, True),       # 64-channel tile (MT = 2)
    (2, 64, 128, 19, 35, False),     # 128-channel tile (MT = 4)
    (1, 32, 128, 204, 36, True),     # a tall map: many row tiles, two column tiles
    (1, 20, 70, 11, 17, True),       # ragged channel counts (zero-padded chunk, partial channel tile, partial output plane)
    (1, 256, 512, 9, 83, True),      # W = 83 as in the 1333x800 block5 map
    (2, 32, 128, 3, 333, False),     # right-edge tiles straddling the image edge; wgrad edge column in the main stage
    (1, 40, 130, 1, 70, True),       # a single row
    (3, 32, 64, 4, 3, True),         # narrower than one 16-B piece
])
def test_conv3x3_bf16_native(ops, n, cin, cout, h, w, relu):
    """ops.conv3x3 in "bf16" mode = the bf16-storage kernels (ptmi_p8_conv3x3 / ptmi_p8_wgrad, v_mfma_f32_32x32x16_bf16) behind
    fp32 NCHW conversions: forward, dgrad, wgrad and bias gradient against torch CPU fp32 on bf16-rounded operands (a bf16 x bf16
    product is exact in fp32: the accumulators differ in summation order only; stored activations / activation gradients are
    additionally rounded to bf16, weight and bias gradients are fp32)."""
    gen = g(n * 1000 + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    xr, wr, br = _rb(x).requires_grad_(), _rb(wt).requires_grad_(), b.clone().requires_grad_()
    zr = F.conv2d(xr, wr, br, padding=1)
    yr = F.relu(zr) if relu else zr
    gy = torch.randn(yr.shape, generator=gen)
    # the gradient that reaches the convolution (after the ReLU mask) is what gets rounded
    gz = _rb(gy * (zr.detach() > 0)) if relu else _rb(gy)
    zr.backward(gz)
    ops.set_operand_rounding("bf16")
    try:
        xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, wt, b))
        yd = ops.conv3x3(xd, wd, bd, relu)
        yd.backward(gy.to(DEV))
    finally:
        ops.set_operand_rounding(None)
    close_stored_bf16(yd, yr, "bf16 conv fwd")
    # ReLU-mask flips where the pre-activation is within rounding of 0 change which gradient elements exist
    stable = (zr.detach().abs() > 1e-4) if relu else torch.ones_like(zr, dtype=torch.bool)
    if bool(stable.all()):
        close_stored_bf16(xd.grad, xr.grad, "bf16 conv dgrad")
        close(wd.grad, wr.grad, 2e-4, 1e-3, "bf16 conv wgrad")
        close(bd.grad, br.grad, 1e-4, 1e-3, "bf16 conv bias grad")
    else:
        assert float(stable.float().mean()) > 0.999


@pytest.mark.parametrize("pool", [False, True])
def test_vgg_block_bf16_native(ops, pool):
    """The VGG block on the bf16-storage kernels (epilogues 1 / 3, P8 pool backward; p8._Block) against the same block run layer
    by layer in "bf16_emulate" mode (separate rounding passes + the fp32 kernels), and the inference-only conv+ReLU+pool."""
    gen = g(197)
    x = torch.randn(2, 16, 22, 37, generator=gen)
    ws = [torch.randn(24, 16, 3, 3, generator=gen) * 0.1, torch.randn(24, 24, 3, 3, generator=gen) * 0.08,
          torch.randn(32, 24, 3, 3, generator=gen) * 0.08]
    bs = [torch.randn(c, generator=gen) * 0.1 for c in (24, 24, 32)]
    gy = None
    res = {}
    for mode in ("bf16_emulate", "bf16"):
        ops.set_operand_rounding(mode)
        try:
            xd = x.to(DEV).requires_grad_()
            wd = [t.to(DEV).requires_grad_() for t in ws]
            bd = [t.to(DEV).requires_grad_() for t in bs]
            yd = ops.vgg_block(xd, pool, [t for pair in zip(wd, bd) for t in pair])
            if gy is None:
                gy = torch.randn(yd.shape, generator=gen).to(DEV)
            yd.backward(gy)
            fused = ops.conv3x3_relu_pool_nograd(x.to(DEV), ws[0].to(DEV), bs[0].to(DEV))
            # 3 input channels (one zero-padded 16-channel chunk in native mode)
            fused3 = ops.conv3x3_relu_pool_nograd(x[:, :3].contiguous().to(DEV), ws[0][:, :3].contiguous().to(DEV), bs[0].to(DEV))
        finally:
            ops.set_operand_rounding(None)
        res[mode] = (yd.detach(), xd.grad, [t.grad for t in wd], [t.grad for t in bd], fused, fused3)
    e, nat = res["bf16_emulate"], res["bf16"]
    # both modes round what is stored (native: in the kernels' epilogues; emulate: by passes): equal up to the summation order of the
    # fp32 accumulators, i.e. up to a bf16 rounding boundary crossed here and there
    close_stored_bf16(nat[0], e[0], "block fwd")
    close_stored_bf16(nat[4], e[4], "fused conv+relu+pool")
    close_stored_bf16(nat[5], e[5], "fused conv+relu+pool, 3 input channels")
    close(nat[1], e[1], 1e-2, 5e-3, "block dx")        # (a handful of ReLU-mask / pool-tie flips between the two summation orders)
    for i in range(3):
        close(nat[2][i], e[2][i], 5e-3, 5e-3, f"block dw{i}")
        close(nat[3][i], e[3][i], 5e-3, 5e-3, f"block db{i}")


def test_conv3x3_bf16_random_shapes(ops):
    """The shape fuzz of the fp32 kernels, on the bf16 entry points (forward + wgrad; no ReLU so no mask flips)."""
    rng = np.random.RandomState(4048)
    ops.set_operand_rounding("bf16")
    try:
        for case in range(16):
            n = int(rng.randint(1, 4))
            cin = int(rng.choice([1, 3, 4, 5, 17, 32, 33, 64]))
            cout = int(rng.choice([1, 7, 32, 64, 65, 128, 130]))
            h = int(rng.choice([1, 2, 3, 4, 5, 9, 13]))
            w = int(rng.choice([1, 2, 3, 5, 8, 25, 31, 32, 33, 39, 57, 64, 71]))
            gen = g(3000 + case)
            x = torch.randn(n, cin, h, w, generator=gen)
            wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
            b = torch.randn(cout, generator=gen) * 0.1
            xr, wr, br = _rb(x).requires_grad_(), _rb(wt).requires_grad_(), b.clone().requires_grad_()
            yr = F.conv2d(xr, wr, br, padding=1)
            gy = torch.randn(yr.shape, generator=gen)
            yr.backward(_rb(gy))
            xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, wt, b))
            yd = ops.conv3x3(xd, wd, bd, False)
            yd.backward(gy.to(DEV))
            tag = f"case {case} n={n} cin={cin} cout={cout} h={h} w={w}"
            close_stored_bf16(yd, yr, "fwd " + tag)
            close_stored_bf16(xd.grad, xr.grad, "dgrad " + tag)
            close(wd.grad, wr.grad, 2e-4, 1e-3, "wgrad " + tag)
            close(bd.grad, br.grad, 1e-4, 1e-3, "bias grad " + tag)
    finally:
        ops.set_operand_rounding(None)


@pytest.mark.parametrize("mode", ["bf16", "bf16_emulate"])
def test_bf16_operand_rounding_mode(ops, mode):
    """SOLVER.AMP.ENABLED (BASELINE configs[4] numerics): conv / FC operands and incoming gradients rounded to bf16, fp32
    accumulation, fp32 results == the fp32 op applied to bf16-rounded tensors."""
    gen = g(99)
    rb = lambda t: t.to(torch.bfloat16).float()
    x = torch.randn(2, 24, 17, 29, generator=gen)
    wt = torch.randn(40, 24, 3, 3, generator=gen) * 0.1
    b = torch.randn(40, generator=gen) * 0.1
    gy = torch.randn(2, 40, 17, 29, generator=gen)
    xr, wr, br = rb(x).requires_grad_(), rb(wt).requires_grad_(), b.clone().requires_grad_()
    yr = F.conv2d(xr, wr, br, padding=1)
    yr.backward(rb(gy))
    lx = torch.randn(70, 300, generator=gen)
    lw = torch.randn(50, 300, generator=gen) * 0.05
    lb = torch.randn(50, generator=gen)
    lg = torch.randn(70, 50, generator=gen)
    lxr, lwr = rb(lx).requires_grad_(), rb(lw).requires_grad_()
    lyr = F.linear(lxr, lwr, lb)
    lyr.backward(rb(lg))
    ops.set_operand_rounding(mode)
    try:
        xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, wt, b))
        yd = ops.conv3x3(xd, wd, bd, False)
        yd.backward(gy.to(DEV))
        lxd, lwd, lbd = (t.to(DEV).requires_grad_() for t in (lx, lw, lb))
        lyd = ops.linear(lxd, lwd, lbd, False)
        lyd.backward(lg.to(DEV))
    finally:
        ops.set_operand_rounding(None)
    close_stored_bf16(yd, yr, "bf16-mode conv fwd")          # stored in bf16 (p8.hip; emulated by a rounding pass)
    close_stored_bf16(xd.grad, xr.grad, "bf16-mode conv dgrad")
    close(wd.grad, wr.grad, 1e-4, 1e-3, "bf16-mode conv wgrad")
    close(lyd, lyr, 1e-4, 1e-4, "bf16-mode linear fwd")
    close(lxd.grad, lxr.grad, 1e-4, 1e-4, "bf16-mode linear dx")
    close(lwd.grad, lwr.grad, 1e-4, 1e-3, "bf16-mode linear dw")
    # and it really is a different result than fp32 (the rounding is applied)
    y32 = ops.conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), False)
    assert float((y32 - yd.detach()).abs().max()) > 1e-3
    with pytest.raises(ValueError):
        ops.set_operand_rounding("fp8")
