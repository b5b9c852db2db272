"""bf16 STORAGE path of the convolution stack (SOLVER.AMP.ENABLED; csrc/p8.hip, C ABI ptmi_p8_*).

A P8 tensor is a bf16 activation (or activation-gradient) tensor in the padded 8-channel-block layout
`t[2 ceil(C/16)][N (H + 1) + 1][W + 1][8]` (include/ptmi355.h).  torch carries the storage (a `torch.bfloat16` tensor), the
current stream and the autograd tape; every operator body is a HIP kernel.  `ops` routes its 3x3-convolution entry points here
when the operand-rounding mode is "bf16" (what SOLVER.AMP.ENABLED selects); the backbone keeps P8 tensors between its layers
(`VGGPipeline`) and converts to fp32 NCHW once, at the feature map the RPN / ROI heads consume.

Numerics (what the parity tests state): a P8 tensor holds values ROUNDED to bf16 (nearest even) at the moment they are
stored; convolutions multiply bf16 operands exactly and accumulate in fp32; weight / bias gradients and everything outside the
convolution stack stay fp32 -- the numerics of ops' "bf16_emulate" mode (fp32 kernels on tensors rounded by separate passes) up
to fp32 summation order and the rounding of the stored results."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from . import ops

BF16 = torch.bfloat16
F32 = torch.float32


def planes(c: int) -> int:
    return 2 * (-(-c // 16))


def plane_pixels(n: int, h: int, w: int) -> int:
    return (n * (h + 1) + 1) * (w + 1)


def _alloc(c: int, n: int, h: int, w: int, device) -> torch.Tensor:
    return torch.empty((planes(c), n * (h + 1) + 1, w + 1, 8), dtype=BF16, device=device)


def _chk(t: torch.Tensor, c: int, n: int, h: int, w: int, name: str):
    want = (planes(c), n * (h + 1) + 1, w + 1, 8)
    if not t.is_cuda or t.dtype != BF16 or tuple(t.shape) != want or not t.is_contiguous():
        raise _lib.PtmiError(f"{name}: expected a contiguous ROCm bf16 P8 tensor of shape {want}, got {t.dtype} {tuple(t.shape)}")
    return t


# ============================================================================ raw launches
def from_nchw(x: torch.Tensor) -> torch.Tensor:
    """fp32 (N, C, H, W) -> P8 storage (values rounded to bf16, channels padded to whole 16-channel chunks with zeros)"""
    x = ops._chk(x.contiguous(), name="p8.from_nchw input")
    n, c, h, w = x.shape
    y = _alloc(c, n, h, w, x.device)
    with ops._prof("p8_convert"):
        _lib.call("ptmi_p8_from_nchw", ops._ptr(x), ops._ptr(y), n, c, h, w, ops._stream())
    return y


def to_nchw(t: torch.Tensor, n: int, c: int, h: int, w: int) -> torch.Tensor:
    y = torch.empty((n, c, h, w), dtype=F32, device=t.device)
    with ops._prof("p8_convert"):
        _lib.call("ptmi_p8_to_nchw", ops._ptr(_chk(t, c, n, h, w, "p8.to_nchw")), ops._ptr(y), n, c, h, w, ops._stream())
    return y


def pack_weights(w: torch.Tensor, mode: int) -> torch.Tensor:
    """fp32 (Cout, Cin, 3, 3) -> bf16 MFMA operand slabs for conv3x3 (mode 0) / its dgrad (mode 1)"""
    w = ops._chk(w.contiguous(), name="conv weight")
    co, ci = w.shape[0], w.shape[1]
    conv_cin, conv_cout = (ci, co) if mode == 0 else (co, ci)
    wp = torch.empty(_lib.load().ptmi_p8_packed_elems(conv_cin, conv_cout), dtype=BF16, device=w.device)
    _lib.call("ptmi_p8_pack_weights", ops._ptr(w), ops._ptr(wp), co, ci, mode, ops._stream())
    return wp


def conv3x3_raw(x: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], mask_ref: Optional[torch.Tensor], n: int,
                cin: int, cout: int, h: int, w: int, epilogue: int) -> torch.Tensor:
    """one launch of ptmi_p8_conv3x3 (epilogue 0: + bias, 1: + bias + ReLU, 2: none, 3: times (mask_ref > 0), 4: + bias + ReLU + 2x2
    max-pool: the result is the P8 tensor of the pooled map)"""
    y = _alloc(cout, n, h // 2, w // 2, x.device) if epilogue == 4 else _alloc(cout, n, h, w, x.device)
    flops = 2.0 * 9 * cin * cout * h * w * n
    nbytes = 2.0 * plane_pixels(n, h, w) * (cin + cout * (2 if epilogue == 3 else (0.25 if epilogue == 4 else 1))) + 2.0 * 9 * cin * cout
    if mask_ref is not None:
        _chk(mask_ref, cout, n, h, w, "p8 conv mask")
    with ops._prof("p8_conv3x3", flops, nbytes):
        _lib.call("ptmi_p8_conv3x3_waves", ops._ptr(_chk(x, cin, n, h, w, "p8 conv input")), ops._ptr(wp),
                  ops._ptr(bias), ops._ptr(mask_ref), ops._ptr(y), n, cin, cout, h, w, epilogue, ops._P8_CONV_WAVES, ops._stream())
    return y


_WGRAD_FALLBACK_BYTES = 1 << 30     # per fp32 operand copy of p8.wgrad's oversize fallback


def _widen_images(t: torch.Tensor, n: int, c: int, h: int, w: int, i0: int, i1: int) -> torch.Tensor:
    """fp32 (i1 - i0, C, H, W) of images i0 .. i1 - 1 of a P8 tensor (exact: bf16 -> fp32).  A strided view + one copy: the
    oversize fallback of wgrad only (the whole-tensor conversion is the HIP kernel behind to_nchw)."""
    v = t[:, 1:, 1:, :].reshape(t.shape[0], n, h + 1, w, 8)[:, i0:i1, :h]          # (planes, k, H, W, 8)
    return v.permute(1, 0, 4, 2, 3).reshape(i1 - i0, t.shape[0] * 8, h, w)[:, :c].to(F32).contiguous()


def wgrad(x: torch.Tensor, dy: torch.Tensor, n: int, cin: int, cout: int, h: int, w: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(dW (cout, cin, 3, 3), db (cout,)) fp32 from the layer's P8 input and P8 output gradient"""
    dw = torch.empty((cout, cin, 3, 3), dtype=F32, device=x.device)
    db = torch.empty(cout, dtype=F32, device=x.device)
    if not _lib.load().ptmi_p8_wgrad_fits(n, cin, cout, h, w):
        # a P8 tensor beyond the kernel's 32-bit offsets (conv1_2 with >= 32 images when blocks 1-2 are trainable): the
        # direct fp32 split-K kernel on the WIDENED operands -- the same bf16-rounded products, fp32 accumulation, another summation
        # order -- over groups of images (accumulate = 1 from the second group on), so that the two fp32 copies stay bounded
        # (8.7 GB each for conv1_2 with 32 images when widened whole)
        _chk(x, cin, n, h, w, "p8 wgrad input")
        _chk(dy, cout, n, h, w, "p8 wgrad grad")
        group = max(1, min(n, _WGRAD_FALLBACK_BYTES // (4 * max(cin, cout) * h * w)))
        for i0 in range(0, n, group):
            i1 = min(n, i0 + group)
            x32, dy32 = _widen_images(x, n, cin, h, w, i0, i1), _widen_images(dy, n, cout, h, w, i0, i1)
            ws = ops._ws("wgrad", _lib.load().ptmi_conv3x3_wgrad_ws_floats(i1 - i0, cin, cout, h, w) * 4, x.device)
            with ops._prof("conv3x3_wgrad", 2.0 * 9 * cin * cout * h * w * (i1 - i0)):
                _lib.call("ptmi_conv3x3_wgrad", ops._ptr(x32), ops._ptr(dy32), ops._ptr(dw), ops._ptr(db), ops._ptr(ws),
                          i1 - i0, cin, cout, h, w, int(i0 > 0), ops._stream())       # (accumulate adds into dw AND db)
        return dw, db
    ws = ops._ws("p8wgrad", _lib.load().ptmi_p8_wgrad_ws_floats_waves(n, cin, cout, h, w, ops._WGRAD_WAVES) * 4, x.device)
    with ops._prof("p8_wgrad", 2.0 * 9 * cin * cout * h * w * n):
        _lib.call("ptmi_p8_wgrad_waves", ops._ptr(_chk(x, cin, n, h, w, "p8 wgrad input")), ops._ptr(_chk(dy, cout, n, h, w, "p8 wgrad grad")),
                  ops._ptr(dw), ops._ptr(db), ops._ptr(ws), n, cin, cout, h, w, 0, ops._WGRAD_WAVES, ops._stream())
    return dw, db


def maxpool_fwd(x: torch.Tensor, n: int, c: int, h: int, w: int) -> torch.Tensor:
    y = _alloc(c, n, h // 2, w // 2, x.device)
    with ops._prof("p8_maxpool_fwd"):
        _lib.call("ptmi_p8_maxpool2x2_fwd", ops._ptr(_chk(x, c, n, h, w, "p8 pool input")), ops._ptr(y), n, c, h, w, ops._stream())
    return y


def maxpool_bwd(x: torch.Tensor, dy: torch.Tensor, n: int, c: int, h: int, w: int, relu_mask: bool) -> torch.Tensor:
    dx = torch.empty_like(x)
    with ops._prof("p8_maxpool_bwd"):
        _lib.call("ptmi_p8_maxpool2x2_bwd", ops._ptr(_chk(x, c, n, h, w, "p8 pool input")),
                  ops._ptr(_chk(dy, c, n, h // 2, w // 2, "p8 pool grad")), ops._ptr(dx), n, c, h, w, int(relu_mask), ops._stream())
    return dx


def relu_bwd(dy: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    dz = torch.empty_like(dy)
    with ops._prof("p8_relu_bwd"):
        _lib.call("ptmi_p8_relu_bwd", ops._ptr(dy), ops._ptr(y), ops._ptr(dz), dy.numel() // 8, ops._stream())
    return dz



# ============================================================================ autograd nodes
class _ToNCHW(torch.autograd.Function):
    """P8 -> fp32 NCHW (exact); backward rounds the fp32 gradient into a P8 gradient"""

    @staticmethod
    def forward(ctx, t, n, c, h, w):
        ctx.meta = (n, c, h, w)
        return to_nchw(t, n, c, h, w)

    @staticmethod
    def backward(ctx, dy):
        return from_nchw(dy), None, None, None, None


class _FromNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.meta = tuple(x.shape)
        return from_nchw(x)

    @staticmethod
    def backward(ctx, dt):
        return to_nchw(dt.contiguous(), *ctx.meta)


class _Block(torch.autograd.Function):
    """k x [conv3x3 + bias + ReLU] (+ MaxPool 2x2) on P8 tensors as ONE autograd node (vgg.py:65-72): the counterpart of
    ops._VGGBlock -- pool backward applies the last ReLU's mask, every dgrad launch the mask of the producing layer (epilogue 3),
    weight + bias gradients come out of one kernel per layer."""

    @staticmethod
    def forward(ctx, x, n, cin, h, w, pool, *wb):
        k = len(wb) // 2
        acts, ws, c = [_chk(x, cin, n, h, w, "block input")], [], cin
        for j in range(k):
            wt, b = ops._chk(wb[2 * j].contiguous()), ops._chk(wb[2 * j + 1].contiguous())
            ws.append(wt)
            acts.append(conv3x3_raw(acts[-1], pack_weights(wt, 0), b, None, n, c, wt.shape[0], h, w, 1))
            c = wt.shape[0]
        out = maxpool_fwd(acts[-1], n, c, h, w) if pool else acts[-1]
        ctx.meta = (k, pool, n, cin, h, w)
        ctx.save_for_backward(*acts, *ws)
        return out

    @staticmethod
    def backward(ctx, dout):
        k, pool, n, cin, h, w = ctx.meta
        saved = ctx.saved_tensors
        acts, ws = saved[: k + 1], saved[k + 1:]
        cout = ws[-1].shape[0]
        dout = dout.contiguous()
        dz = maxpool_bwd(acts[k], dout, n, cout, h, w, True) if pool else relu_bwd(dout, acts[k])
        grads = [None] * (2 * k)
        dx = None
        for j in range(k, 0, -1):
            xin, wt = acts[j - 1], ws[j - 1]
            co, ci = wt.shape[0], wt.shape[1]
            if ctx.needs_input_grad[6 + 2 * (j - 1)] or ctx.needs_input_grad[7 + 2 * (j - 1)]:
                grads[2 * (j - 1)], grads[2 * (j - 1) + 1] = wgrad(xin, dz, n, ci, co, h, w)
            if j > 1:
                dz = conv3x3_raw(dz, pack_weights(wt, 1), None, xin, n, co, ci, h, w, 3)       # dgrad + ReLU mask of layer j - 1
            elif ctx.needs_input_grad[0]:
                dx = conv3x3_raw(dz, pack_weights(wt, 1), None, None, n, co, ci, h, w, 2)
        return (dx, None, None, None, None, None, *grads)


def block(x: torch.Tensor, n: int, cin: int, h: int, w: int, pool: bool, params) -> torch.Tensor:
    """P8 in, P8 out.  params = [w1, b1, w2, b2, ...]"""
    return _Block.apply(x, n, cin, h, w, pool, *params)


# ---------------------------------------------------------------------------- fp32-NCHW-facing entry points (ops routes here)
def conv3x3_nchw(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, relu: bool) -> torch.Tensor:
    """ops.conv3x3 in "bf16" mode: fp32 NCHW in and out, the convolution and its gradients on the P8 kernels"""
    n, cin, h, w = x.shape
    t = _FromNCHW.apply(x) if x.requires_grad else from_nchw(x)
    y = _Conv.apply(t, n, cin, h, w, relu, weight, bias)
    return _ToNCHW.apply(y, n, weight.shape[0], h, w)


class _Conv(torch.autograd.Function):
    """one conv3x3 + bias (+ ReLU) on P8 tensors"""

    @staticmethod
    def forward(ctx, x, n, cin, h, w, relu, weight, bias):
        weight, bias = ops._chk(weight.contiguous(), name="conv weight"), ops._chk(bias.contiguous(), name="conv bias")
        y = conv3x3_raw(x, pack_weights(weight, 0), bias, None, n, cin, weight.shape[0], h, w, 1 if relu else 0)
        ctx.meta = (n, cin, h, w, relu)
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, cin, h, w, relu = ctx.meta
        x, weight, y = ctx.saved_tensors
        cout = weight.shape[0]
        dz = relu_bwd(dy.contiguous(), y) if relu else dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[6] or ctx.needs_input_grad[7]:
            dw, db = wgrad(x, dz, n, cin, cout, h, w)
        if ctx.needs_input_grad[0]:
            dx = conv3x3_raw(dz, pack_weights(weight, 1), None, None, n, cout, cin, h, w, 2)
        return dx, None, None, None, None, None, dw, db


def vgg_block_nchw(x: torch.Tensor, pool: bool, params) -> torch.Tensor:
    """ops.vgg_block in "bf16" mode (fp32 NCHW in and out)"""
    n, cin, h, w = x.shape
    t = _FromNCHW.apply(x) if x.requires_grad else from_nchw(x)
    y = block(t, n, cin, h, w, pool, params)
    cout = params[-2].shape[0]
    return _ToNCHW.apply(y, n, cout, h // 2 if pool else h, w // 2 if pool else w)


def conv3x3_relu_pool_nograd_nchw(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    y = conv3x3_raw(from_nchw(x), pack_weights(weight, 0), ops._chk(bias.contiguous()), None, n, cin, cout, h, w, 4)
    return to_nchw(y, n, cout, h // 2, w // 2)


# ============================================================================ bf16-storage GEMM (the box head's large Linear layer)
def pack_matrix(src: torch.Tensor, rows: int, k: int, ld: int, k_major: bool) -> torch.Tensor:
    """fp32 matrix -> bf16 "P8 matrix" operand t[ceil(k/8)][rows][8]: element (row, kk) = src[row * ld + kk] (k_major) or
    src[kk * ld + row] (not k_major)"""
    dst = torch.empty((-(-k // 8), rows, 8), dtype=BF16, device=src.device)
    with ops._prof("p8m_pack"):
        _lib.call("ptmi_p8m_pack", ops._ptr(ops._chk(src, name="p8m_pack source")), ops._ptr(dst), rows, k, ld, int(k_major), ops._stream())
    return dst


def gemm_nt(a: torch.Tensor, b: torch.Tensor, m: int, n: int, k: int, bias: Optional[torch.Tensor] = None, relu: bool = False,
            swapped: bool = False) -> torch.Tensor:
    """C (m, n) fp32 = A . B^T (+ bias) (+ ReLU) from two packed operands (pack_matrix) with k octets each.
    swapped: compute C^T = B . A^T and store it transposed -- the same matrix through 16-byte stores (large, store-bound outputs)"""
    if not _lib.load().ptmi_p8_gemm_nt_fits(m, n, k):
        raise _lib.PtmiError(f"p8.gemm_nt: operands of {m} x {k} / {n} x {k} exceed the kernel's 32-bit buffer offsets (4 GiB per packed "
                             "operand: fc1 up to ~85 k ROIs per call) -- lower the per-GPU batch or SOLVER.AMP.ENABLED = False")
    c = torch.empty((m, n), dtype=F32, device=a.device)
    if swapped:
        assert bias is None and not relu
        with ops._prof("p8_gemm", 2.0 * m * n * k):
            _lib.call("ptmi_p8_gemm_nt", ops._ptr(b), ops._ptr(a), ops._ptr(c), None, None, n, m, k, n, 2, ops._stream())
        return c
    nws = _lib.load().ptmi_p8_gemm_nt_ws_floats(m, n, k)
    ws = ops._ws("p8gemm", nws * 4, a.device) if nws else None
    with ops._prof("p8_gemm", 2.0 * m * n * k):
        _lib.call("ptmi_p8_gemm_nt", ops._ptr(a), ops._ptr(b), ops._ptr(c), ops._ptr(bias), ops._ptr(ws), m, n, k, n, int(relu), ops._stream())
    return c


class _LinearP8(torch.autograd.Function):
    """y = x W^T + b (+ ReLU) with bf16-storage operands: forward, dX and dW are three launches of ONE kernel (C = A . B^T), each on
    operands packed with their contraction index as k; fp32 accumulation, fp32 y / dX / dW / db."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu: bool, recording: bool = True):
        x = ops._chk(x.contiguous(), name="linear input")
        weight, bias = ops._chk(weight.contiguous()), ops._chk(bias.contiguous())
        r, k = x.shape
        n = weight.shape[0]
        y = gemm_nt(pack_matrix(x, r, k, k, True), pack_matrix(weight, n, k, k, True), r, n, k, bias, relu)
        # for dW the input is needed with the ROW index as k: packed now (bf16: half of what saving x would hold)
        # (`recording` = torch.is_grad_enabled() at the CALL -- inside forward() grad mode is always off -- : inference with trainable
        # weights must not pack and write a second R x 25088 bf16 tensor nobody reads)
        xt = pack_matrix(x, k, r, k, False) if (recording and ctx.needs_input_grad[1]) else None
        ctx.meta = (r, k, n, relu, ctx.needs_input_grad[0])
        ctx.save_for_backward(xt, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xt, weight, y = ctx.saved_tensors
        r, k, n, relu, need_dx = ctx.meta
        dy = ops._chk(dy.contiguous())
        dz = ops.relu_bwd(dy, y) if relu else dy
        dx = dw = db = None
        if need_dx:                                        # dX = dZ W:  A = dZ (k = n), B = W^T (rows = input feature, k = n)
            dx = gemm_nt(pack_matrix(dz, r, n, n, True), pack_matrix(weight, k, n, k, False), r, k, n, swapped=True)
        if ctx.needs_input_grad[1]:                        # dW = dZ^T X:  A = dZ^T (rows n, k = r), B = X^T (rows = input feature, k = r)
            dw = gemm_nt(pack_matrix(dz, n, r, n, False), xt, n, k, r)
        if ctx.needs_input_grad[2]:                        # the bias gradient sums the values the GEMMs consumed (rounded)
            db = ops.colsum(dz.to(BF16).to(F32))
        return dx, dw, db, None, None


class _RoiAlignLinearP8(torch.autograd.Function):
    """ROIPooler -> flatten -> Linear (+ ReLU) of the box head (roi_heads.py:126-128 -> FastRCNNConvFCHead.fc1) as ONE node under
    SOLVER.AMP.ENABLED: the ROIAlign kernel writes the Linear layer's bf16 operands itself (no fp32 ROI-feature tensor, no pack passes
    over it); backward = _LinearP8's three GEMMs, then the grouped ROIAlign backward on dX."""

    @staticmethod
    def forward(ctx, feat, rois, img_offsets, pooled: int, scale: float, weight, bias, relu: bool, recording: bool = True):
        weight, bias = ops._chk(weight.contiguous()), ops._chk(bias.contiguous())
        n_img, c, h, w = feat.shape
        r, k, n = rois.shape[0], c * pooled * pooled, weight.shape[0]
        need_w = recording and ctx.needs_input_grad[5]       # (see _LinearP8.forward)
        xk, xt = ops.roi_align_p8m(feat, rois, img_offsets, pooled, scale, need_w)
        y = gemm_nt(xk, pack_matrix(weight, n, k, k, True), r, n, k, bias, relu)
        ctx.meta = (r, k, n, relu, (n_img, c, h, w), pooled, float(scale))
        ctx.save_for_backward(xt, weight, y if relu else None, rois, img_offsets)
        return y

    @staticmethod
    def backward(ctx, dy):
        xt, weight, y, rois, img_offsets = ctx.saved_tensors
        r, k, n, relu, (n_img, c, h, w), pooled, scale = ctx.meta
        dy = ops._chk(dy.contiguous())
        dz = ops.relu_bwd(dy, y) if relu else dy
        dfeat = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm_nt(pack_matrix(dz, r, n, n, True), pack_matrix(weight, k, n, k, False), r, k, n, swapped=True)
            dfeat = ops.roi_align_bwd_grouped(dx, rois, img_offsets, n_img, c, h, w, pooled, scale)
        if ctx.needs_input_grad[5]:
            dw = gemm_nt(pack_matrix(dz, n, r, n, False), xt, n, k, r)
        if ctx.needs_input_grad[6]:
            db = ops.colsum(dz.to(BF16).to(F32))
        return dfeat, None, None, None, None, dw, db, None, None


def roi_align_linear(feat, rois, img_offsets, pooled: int, scale: float, weight, bias, relu: bool) -> torch.Tensor:
    return _RoiAlignLinearP8.apply(feat, rois, img_offsets, pooled, scale, weight, bias, relu, torch.is_grad_enabled())


LINEAR_MIN_K = 4096        # below this (fc2, the predictors) the operand packs cost what the GEMM saves: ptmi_gemm_bf16 keeps those


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, relu: bool) -> torch.Tensor:
    return _LinearP8.apply(x, weight, bias, relu, torch.is_grad_enabled())
