"""bf16 STORAGE path of the convolution stack (SOLVER.AMP.ENABLED; csrc/p8.hip, C ABI ptmi_p8_*).

A `P8` value is a bf16 activation (or activation-gradient) tensor in the padded 8-channel-block layout
`t[ceil(C/8)][N (H + 1) + 1][W + 1][8]` (include/ptmi355.h) together with its logical shape.  torch carries the storage
(a `torch.bfloat16` tensor), the current stream and the autograd tape; every operator body is a HIP kernel.

Numerics (what the parity tests state): a P8 tensor holds values ROUNDED to bf16 (nearest even) at the moment they are
stored; convolutions multiply bf16 operands exactly and accumulate in fp32 -- the same numbers, up to fp32 summation order,
as ops' "bf16" operand-rounding mode produces with fp32 tensors in HBM (a rounded activation is what the next layer
consumes either way), at half the bytes."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from . import ops

BF16 = torch.bfloat16
F32 = torch.float32


def plane_pixels(n: int, h: int, w: int) -> int:
    return (n * (h + 1) + 1) * (w + 1)


def _alloc(cb: int, n: int, h: int, w: int, device) -> torch.Tensor:
    return torch.empty((cb, n * (h + 1) + 1, w + 1, 8), dtype=BF16, device=device)


def _chk(t: torch.Tensor, c: int, n: int, h: int, w: int, name: str):
    want = (-(-c // 8), n * (h + 1) + 1, w + 1, 8)
    if not t.is_cuda or t.dtype != BF16 or tuple(t.shape) != want or not t.is_contiguous():
        raise _lib.PtmiError(f"{name}: expected a contiguous ROCm bf16 P8 tensor of shape {want}, got {t.dtype} {tuple(t.shape)}")
    return t


def from_nchw(x: torch.Tensor, cb_out: Optional[int] = None) -> torch.Tensor:
    """fp32 (N, C, H, W) -> P8 storage with cb_out (default ceil(C/8)) channel blocks; values rounded to bf16"""
    x = ops._chk(x.contiguous(), name="p8.from_nchw input")
    n, c, h, w = x.shape
    cb = -(-c // 8) if cb_out is None else cb_out
    y = _alloc(cb, n, h, w, x.device)
    with ops._prof("p8_convert"):
        _lib.call("ptmi_p8_from_nchw", ops._ptr(x), ops._ptr(y), n, c, cb, h, w, ops._stream())
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
    """one launch of ptmi_p8_conv3x3; cin = the channel count the P8 input is padded to (a multiple of 16)"""
    y = _alloc(cout // 8, n, h, w, x.device)
    flops = 2.0 * 9 * cin * cout * h * w * n
    nbytes = 2.0 * plane_pixels(n, h, w) * (cin + cout * (2 if epilogue == 3 else 1)) + 2.0 * 9 * cin * cout
    with ops._prof("p8_conv3x3", flops, nbytes):
        _lib.call("ptmi_p8_conv3x3", ops._ptr(_chk(x, cin, n, h, w, "p8 conv input")), ops._ptr(wp),
                  ops._ptr(bias), ops._ptr(mask_ref), ops._ptr(y), n, cin, cout, h, w, epilogue, ops._stream())
    return y


def maxpool_fwd(x: torch.Tensor, n: int, c: int, h: int, w: int) -> torch.Tensor:
    y = _alloc(c // 8, n, h // 2, w // 2, x.device)
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
    _lib.call("ptmi_p8_relu_bwd", ops._ptr(dy), ops._ptr(y), ops._ptr(dz), dy.numel() // 8, ops._stream())
    return dz


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(a)
    _lib.call("ptmi_p8_add", ops._ptr(a), ops._ptr(b), ops._ptr(out), a.numel() // 8, ops._stream())
    return out


def wgrad(x: torch.Tensor, dy: torch.Tensor, n: int, cin: int, cout: int, h: int, w: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(dW (cout, cin, 3, 3), db (cout,)) fp32 from the layer's P8 input and P8 output gradient; cin = x's padded channel count"""
    dw = torch.empty((cout, cin, 3, 3), dtype=F32, device=x.device)
    db = torch.empty(cout, dtype=F32, device=x.device)
    ws = ops._ws("p8wgrad", _lib.load().ptmi_p8_wgrad_ws_floats(n, cin, cout, h, w) * 4, x.device)
    with ops._prof("p8_wgrad", 2.0 * 9 * cin * cout * h * w * n):
        _lib.call("ptmi_p8_wgrad", ops._ptr(_chk(x, cin, n, h, w, "p8 wgrad input")), ops._ptr(_chk(dy, cout, n, h, w, "p8 wgrad grad")),
                  ops._ptr(dw), ops._ptr(db), ops._ptr(ws), n, cin, cout, h, w, 0, ops._stream())
    return dw, db
