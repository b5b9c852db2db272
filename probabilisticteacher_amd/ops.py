"""torch-facing wrappers of the HIP operators in libptmi355.so.

PyTorch is used here for device memory (caching allocator), the current HIP stream and the autograd tape
only; every operator body is a hand-written gfx950 kernel reached through the C ABI (include/ptmi355.h).
There is no CPU path: tensors must live on a ROCm device and the shared library must be present.
"""
from __future__ import annotations

import contextlib
import ctypes
import functools
import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib

F32 = torch.float32


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, dtype=F32, name="tensor"):
    if not t.is_cuda:
        raise _lib.PtmiError(f"{name}: expected a ROCm device tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.PtmiError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.PtmiError(f"{name}: expected a contiguous tensor")
    return t


_WS = {}


def _ws(name: str, nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffers, one set per (device, stream): kernels of one stream are ordered, so reuse within it is
    safe; work on another stream (a teacher forward on its own stream, say) gets its own buffers instead of racing."""
    key = (name, device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _loss_ws(device) -> torch.Tensor:
    return _ws("loss", 4096 * 4, device)


# ---------------------------------------------------------------------------- HIP-event profiling (bench.py)
_PROF = None


def profile_start() -> None:
    """Start recording a HIP event pair around every launch group below, on the stream the kernels use."""
    global _PROF
    _PROF = []


def profile_stop() -> dict:
    """Synchronise and return {kernel: {"ms", "flops", "issued", "bytes", "calls"}} accumulated since profile_start()."""
    global _PROF
    rec, _PROF = _PROF or [], None
    torch.cuda.synchronize()
    out = {}
    for name, flops, nbytes, issued, e0, e1 in rec:
        d = out.setdefault(name, {"ms": 0.0, "flops": 0.0, "issued": 0.0, "bytes": 0.0, "calls": 0})
        d["ms"] += e0.elapsed_time(e1)
        d["flops"] += flops
        d["issued"] += issued
        d["bytes"] += nbytes
        d["calls"] += 1
    return out


class _prof:
    """flops / nbytes = ALGORITHMIC work of the launch group (every operand read once, every result written once; for a
    convolution the DIRECT-convolution FLOPs, whatever algorithm runs); issued = FLOPs of the MFMA instructions the launch
    actually issues (Winograd: 16 instead of 36 multiplies per 2x2 tile and channel pair, plus its tile padding)."""

    def __init__(self, name, flops=0.0, nbytes=0.0, issued=None):
        self.name, self.flops, self.nbytes = name, float(flops), float(nbytes)
        self.issued = float(flops if issued is None else issued)

    def __enter__(self):
        if _PROF is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _PROF is not None:
            self.e1.record()
            _PROF.append((self.name, self.flops, self.nbytes, self.issued, self.e0, self.e1))
        return False


# ============================================================================ bf16 operands (SOLVER.AMP.ENABLED)
# BASELINE.json configs[4] ("mixed bf16 convs + fp32 loss"; the reference's AMP flag, pt/engine/trainer.py:98): the
# conv / FC GEMM operands -- activations, weights and, in backward, the incoming gradients -- are rounded to bf16
# (round-to-nearest-even), products are accumulated in fp32; losses, box codec, NMS, optimiser are untouched.  Two ways to run it:
#   "bf16"          what SOLVER.AMP.ENABLED selects.  3x3 convolutions: the bf16-STORAGE kernels (p8.py / csrc/p8.hip, round 4):
#                   activations and activation gradients live in bf16 in HBM and LDS (what autocast stores in the reference), in
#                   the padded 8-channel-block layout; the backbone keeps them that way from the image to the block-5 feature
#                   map.  FC layers / 1x1 convolutions: ptmi_gemm_bf16 (fp32 tensors, operands rounded between LDS and the MFMA).
#   "bf16_emulate"  the same numerics on the fp32 kernels: operands are rounded by a separate pass (x.to(bf16).to(f32))
#                   and multiplied by v_mfma_f32_32x32x2_f32 -- a bf16 x bf16 product is exact in fp32 -- and what the storage
#                   kernels store in bf16 (3x3-conv outputs and input gradients) is rounded by a pass as well (`_rnd_stored`;
#                   it matters beyond the next layer's operand rounding: max-pool backward breaks TIES between rounded values
#                   as autocast's bf16 pooling does), so the two modes differ in summation order only.  The native kernels
#                   are tested against this mode.
# Never enabled by bench.py's headline run (the headline metric is fp32).
_OPERAND_ROUNDING = None


def set_operand_rounding(mode: Optional[str]) -> None:
    global _OPERAND_ROUNDING
    if mode not in (None, "bf16", "bf16_emulate"):
        raise ValueError(f"unknown operand rounding {mode!r}")
    _OPERAND_ROUNDING = mode


@contextlib.contextmanager
def operand_rounding(mode: Optional[str]):
    """Scope the operand-rounding mode: a trainer (or an evaluation) carries its own mode and applies it around its forward /
    backward, so that two models with different SOLVER.AMP settings in one process do not change each other's numerics."""
    prev = _OPERAND_ROUNDING
    set_operand_rounding(mode)
    try:
        yield
    finally:
        set_operand_rounding(prev)


def _rnd(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if _OPERAND_ROUNDING != "bf16_emulate" or t is None:
        return t
    return t.to(torch.bfloat16).to(F32)


def _rnd_stored(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """bf16_emulate: a value the bf16-storage path would keep in bf16 (a 3x3-conv output or input gradient)"""
    return _rnd(t)


def _rnd_grad(t: torch.Tensor) -> torch.Tensor:
    """Incoming gradient of a Linear / 1x1 conv: rounded by a tensor pass in BOTH bf16 modes -- besides the GEMMs (which
    would round it themselves in native mode) it feeds the bias-gradient reductions, which must see the same values."""
    return t if _OPERAND_ROUNDING is None else t.to(torch.bfloat16).to(F32)


def _native_bf16() -> bool:
    return _OPERAND_ROUNDING == "bf16"


def _no_native_bf16(what: str) -> None:
    if _native_bf16():
        raise _lib.PtmiError(f"{what} is an fp32-tensor entry point: in \"bf16\" mode the 3x3 convolutions run on the bf16-storage "
                             "kernels (probabilisticteacher_amd.p8; ops.conv3x3 / ops.vgg_block route there)")


def conv3x3_wgrad(x: torch.Tensor, dz: torch.Tensor, cout: int):
    """(dW, db) of a 3x3 s1 p1 convolution from its input and output gradient (N1 wgrad): the Winograd-domain kernel on
    the fp32 layers with >= 64 input and output channels whose map fits its 32-bit offsets, the direct split-K kernel otherwise
    (and under bf16 operand rounding)."""
    _no_native_bf16("ops.conv3x3_wgrad")
    n, cin, h, w = x.shape
    dw = torch.empty(cout, cin, 3, 3, dtype=F32, device=x.device)
    db = torch.empty(cout, dtype=F32, device=x.device)
    lib = _lib.load()
    kind = _wgrad_kind(cin, cout, h, w)
    if kind == "wino4w":
        # round 5: the F(4x4,3x3)-domain kernel (csrc/wino4w.hip; workgroup = 64 co x 32 ci)
        ws = _ws("wgrad", lib.ptmi_conv3x3_wino4_wgrad_ws_floats_waves(n, cin, cout, h, w, _WGRAD_WAVES) * 4, x.device)
        with _prof("conv3x3_wino4_wgrad", 2.0 * 9 * cin * cout * h * w * n, issued=wino4_wgrad_issued_flops(n, cin, cout, h, w)):
            _lib.call("ptmi_conv3x3_wino4_wgrad_waves", _ptr(x), _ptr(dz), _ptr(dw), _ptr(db), _ptr(ws), n, cin, cout, h, w, 0,
                      _WGRAD_WAVES, _stream())
    elif kind == "wino":
        ws = _ws("wgrad", lib.ptmi_conv3x3_wino_wgrad_ws_floats_waves(n, cin, cout, h, w, _WGRAD_WAVES) * 4, x.device)
        with _prof("conv3x3_wino_wgrad", 2.0 * 9 * cin * cout * h * w * n, issued=wino_wgrad_issued_flops(n, cin, cout, h, w)):
            _lib.call("ptmi_conv3x3_wino_wgrad_waves", _ptr(x), _ptr(dz), _ptr(dw), _ptr(db), _ptr(ws), n, cin, cout, h, w, 0,
                      _WGRAD_WAVES, _stream())
    else:
        ws = _ws("wgrad", lib.ptmi_conv3x3_wgrad_ws_floats(n, cin, cout, h, w) * 4, x.device)
        with _prof("conv3x3_wgrad", 2.0 * 9 * cin * cout * h * w * n):
            _lib.call("ptmi_conv3x3_wgrad", _ptr(x), _ptr(dz), _ptr(dw), _ptr(db), _ptr(ws), n, cin, cout, h, w, 0, _stream())
    return dw, db


# ============================================================================ conv 3x3
# Forward / dgrad algorithm of the fp32 3x3 layers: "auto" = fused Winograd F(4x4,3x3) (csrc/wino4.hip, round 5) where the
# input channel count is a multiple of 8 and >= 64, else fused Winograd F(2x2,3x3) (csrc/wino.hip) wherever the
# input has enough channels to amortise its per-workgroup prologue, the direct implicit GEMM (csrc/conv.hip) for the
# 3-channel stem; "wino2" = as "auto" without the F(4x4,3x3) kernel; "direct" = the direct kernel everywhere (comparison
# runs, tests).  "bf16_emulate" keeps the direct
# kernels (its parity statement -- products of rounded operands are exact in fp32 -- does not survive a transform); "bf16"
# does not come here at all (p8.py).
_CONV_ALGO = "auto"
_WINO_MIN_CIN = 32
_WINO4_MIN_CIN = 64
_WINO_WGRAD_MIN_C = 64         # the wgrad workgroup owns 64 co x 64 ci
_WINO4_WGRAD_MIN_CIN = 32      # the F(4x4,3x3) wgrad workgroup owns 64 co x 32 ci
_WINO4_WGRAD_MIN_FILL = 0.9    # ... and walks the map in chunks of 4 rows x 16 columns: below this share of real pixels per chunk the
                               # F(2x2,3x3)-domain kernel (7- or 8-k-step chunks, whichever fits the row) is faster (measured: 50 x 83,
                               # fill 0.83: 0.91x; 100 x 166, fill 0.94: 1.08x; 200 x 333: 1.17x -- tools/exp/wino4w_bench.py)


def _wino4_wgrad_fill(h: int, w: int) -> float:
    return (h * w) / float(-(-h // 4) * 4 * -(-w // 16) * 16)


def _wgrad_kind(cin: int, cout: int, h: int, w: int) -> str:
    """'wino4w' | 'wino' | 'direct': the kernel family ops.conv3x3_wgrad routes a weight-gradient launch of this shape to
    (F(4x4,3x3)-domain csrc/wino4w.hip, F(2x2,3x3)-domain csrc/wino.hip, direct split-K csrc/conv.hip)"""
    lib = _lib.load()
    if (_CONV_ALGO == "auto" and _OPERAND_ROUNDING is None and cin >= _WINO4_WGRAD_MIN_CIN and cout >= _WINO_WGRAD_MIN_C
            and lib.ptmi_conv3x3_wino4_wgrad_fits(h, w) and _wino4_wgrad_fill(h, w) >= _WINO4_WGRAD_MIN_FILL):
        return "wino4w"
    if (_use_wino(cin) and cin >= _WINO_WGRAD_MIN_C and cout >= _WINO_WGRAD_MIN_C
            and lib.ptmi_conv3x3_wino_wgrad_fits(h, w)):
        return "wino"
    return "direct"


def set_conv_algo(mode: str) -> None:
    global _CONV_ALGO
    if mode not in ("auto", "wino2", "direct"):
        raise ValueError(f"unknown conv algorithm {mode!r}")
    _CONV_ALGO = mode


@functools.lru_cache(maxsize=256)
def wino_issued_flops(n: int, cin: int, cout: int, h: int, w: int) -> float:
    """FLOPs of the v_mfma_f32_32x32x2_f32 instructions one ptmi_conv3x3_wino_fwd launch issues: a wave runs 16 MFMAs
    (4096 FLOP each) per 2-channel k-step for its 32 channels x 32 tiles (2 tile rows x 16 tile columns of the FLAT tile line:
    every (image, 8-row band) is a strip of `period` columns, csrc/wino.hip); waves none of whose tiles lies inside an image
    issue none.  (Checked against SQ_INSTS_VALU_MFMA_MOPS_F32, profiles/r03_*.)"""
    import numpy as np
    co_tiles, chunks, bands = -(-cout // 64), -(-cin // 8), -(-h // 8)
    period = (w + 4) & ~3
    n_strips = n * bands
    n_pix = -(-(n_strips * period) // 32)
    tu = (np.arange(n_pix, dtype=np.int64)[:, None] * 32 + 2 * np.arange(16, dtype=np.int64)[None, :])
    st, px = tu // period, tu % period
    col_ok = (st < n_strips) & (px < w)
    band = st % bands
    waves = 0
    for wn in (0, 1):
        waves += int((col_ok & (band * 8 + 4 * wn < h)).any(axis=1).sum())
    return float(waves) * 2 * co_tiles * chunks * 4 * 16 * 4096


@functools.lru_cache(maxsize=256)
def wino4_issued_flops(n: int, cin: int, cout: int, h: int, w: int) -> float:
    """FLOPs of the v_mfma_f32_16x16x4_f32 instructions one ptmi_conv3x3_wino4_fwd launch issues: a wave runs 72 MFMAs
    (2048 FLOP each: 36 positions x two 16-channel tiles) per 4-channel chunk for its 32 channels x 16 tiles (one tile row of
    the FLAT tile line, csrc/wino4.hip); the persistent kernel runs every wave of every tile (a wave whose tile row lies below
    the image would idle its SIMD either way).  (Checked against SQ_INSTS_VALU_MFMA_MOPS_F32 x 512, profiles/r05_*.)"""
    co_tiles, chunks, bands = -(-cout // 64), cin // 4, -(-h // 8)
    period = (w + 4) & ~3
    n_pix = -(-(n * bands * period) // 64)
    return float(n_pix) * co_tiles * 4 * chunks * 72 * 2048


def wino4_wgrad_issued_flops(n: int, cin: int, cout: int, h: int, w: int) -> float:
    """FLOPs of the MFMAs one ptmi_conv3x3_wino4_wgrad launch issues: per 64 co x 32 ci pair and chunk (one tile row x 16
    columns = 4 tiles = two k-steps) 2 x 18 v_mfma_f32_32x32x2_f32 x 4 waves x 4096 FLOP (csrc/wino4w.hip)"""
    pairs = -(-cout // 64) * -(-cin // 32)
    chunks = n * -(-h // 4) * -(-w // 16)
    return float(pairs) * chunks * 2 * 18 * 4 * 4096


def wino_wgrad_issued_flops(n: int, cin: int, cout: int, h: int, w: int) -> float:
    """FLOPs of the MFMAs one ptmi_conv3x3_wino_wgrad launch issues: per 64 co x 64 ci pair and chunk (one tile row x KSN
    tile pairs; KSN = 7 or 8, whichever pads a tile row less -- csrc/wino.hip: wino_wgrad_ksn) KSN k-steps x 16 positions x
    4 waves x 4096 FLOP"""
    pairs = -(-cout // 64) * -(-cin // 64)
    tile_pairs = -(-(-(-w // 2)) // 2)
    ksn = 7 if -(-tile_pairs // 7) * 7 < -(-tile_pairs // 8) * 8 else 8
    chunks = n * -(-h // 2) * -(-w // (4 * ksn))
    return float(pairs) * chunks * ksn * 16 * 4 * 4096


def _use_wino(conv_cin: int, conv_cout: Optional[int] = None, hw: Optional[Tuple[int, int]] = None) -> bool:
    """Winograd routing of a 3x3 layer: fp32, enough input channels, and -- when the caller knows the map size -- a shape
    that fits the kernel's 32-bit buffer offsets (ptmi_conv3x3_wino_fwd_fits; larger maps run the direct kernel)."""
    if not (_CONV_ALGO in ("auto", "wino2") and _OPERAND_ROUNDING is None and conv_cin >= _WINO_MIN_CIN):
        return False
    if hw is None or conv_cout is None:
        return True
    return bool(_lib.load().ptmi_conv3x3_wino_fwd_fits(conv_cin, conv_cout, int(hw[0]), int(hw[1])))


def _use_wino4(conv_cin: int, conv_cout: Optional[int] = None, hw: Optional[Tuple[int, int]] = None) -> bool:
    """F(4x4,3x3) routing of a 3x3 layer (checked BEFORE _use_wino): fp32, input channels a multiple of 8 and >= 64, and -- when
    the caller knows the map size -- a shape that fits the kernel's 32-bit buffer offsets (ptmi_conv3x3_wino4_fwd_fits)."""
    if not (_CONV_ALGO == "auto" and _OPERAND_ROUNDING is None and conv_cin >= _WINO4_MIN_CIN and conv_cin % 8 == 0):
        return False
    if hw is None or conv_cout is None:
        return True
    return bool(getattr(_lib.load(), _CONV_ABI["wino4"] + "_fwd_fits")(conv_cin, conv_cout, int(hw[0]), int(hw[1])))


def _conv_kind(conv_cin: int, conv_cout: Optional[int] = None, hw: Optional[Tuple[int, int]] = None) -> str:
    """'wino4' | 'wino' | 'mfma' (the direct kernel): the kernel family a forward / dgrad launch of this shape is routed to"""
    if _use_wino4(conv_cin, conv_cout, hw):
        return "wino4"
    return "wino" if _use_wino(conv_cin, conv_cout, hw) else "mfma"


# kind -> C-ABI family.  "wino4" is served by csrc/wino4p.hip since round 6 (positions split over the wave pair: 2 - 4 % faster on every
# layer shape but one, tools/exp/wino4_bench.py --p); set_wino4_family("wino4") selects round 5's csrc/wino4.hip (comparison runs)
_WINO4_FAMILY = "wino4p"


class _ConvAbi(dict):
    def __getitem__(self, kind):
        return "ptmi_conv3x3_" + _WINO4_FAMILY if kind == "wino4" else dict.__getitem__(self, kind)


_CONV_ABI = _ConvAbi({"wino4": "ptmi_conv3x3_wino4p", "wino": "ptmi_conv3x3_wino", "mfma": "ptmi_conv3x3"})


def set_wino4_family(name: str) -> None:
    """the kernel family behind the "wino4" route: "wino4p" (default, csrc/wino4p.hip) | "wino4" (csrc/wino4.hip).  Packed weights
    are family-specific: packs made before the switch must not be used after it."""
    global _WINO4_FAMILY
    if name not in ("wino4", "wino4p"):
        raise ValueError(f"unknown F(4x4,3x3) kernel family {name!r}")
    _WINO4_FAMILY = name


def _conv_issued(kind: str, n: int, cin: int, cout: int, h: int, w: int):
    return (wino4_issued_flops(n, cin, cout, h, w) if kind == "wino4" else
            wino_issued_flops(n, cin, cout, h, w) if kind == "wino" else None)


def conv3x3_pack(w: torch.Tensor, mode: int, epilogue: int, hw: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """Packed weights for conv3x3_raw(..., epilogue) (mode 0) or for the dgrad launch (mode 1: epilogue 2 / 3).  hw = (H, W)
    of the map the weights will be applied to: decides between the Winograd and the direct pack for maps beyond the Winograd
    kernel's 32-bit offsets (None = the map fits; conv3x3_raw checks the pack it is handed against its own routing)."""
    _no_native_bf16("ops.conv3x3_pack")
    _chk(w, name="conv weight")
    co, ci = w.shape[0], w.shape[1]
    conv_cin, conv_cout = (ci, co) if mode == 0 else (co, ci)
    abi = _CONV_ABI[_conv_kind(conv_cin, conv_cout, hw)]
    n = getattr(_lib.load(), abi + "_packed_floats")(conv_cin, conv_cout)
    wp = torch.empty(n, dtype=F32, device=w.device)
    _lib.call(abi + "_pack_weights", _ptr(w), _ptr(wp), co, ci, mode, _stream())
    return wp


_TILE_SCHEDULE = "static"       # persistent F(4x4,3x3) kernel: "dynamic" (work queues, csrc/wino4.hip) | "static" (b, b + grid, ...);
                                # PTrainer selects "dynamic" when the gradient exchange is active
_SCHED_BUFS: dict = {}


def set_tile_schedule(mode: str) -> None:
    """Tile schedule of the persistent forward / dgrad kernel: "static" (the walk b, b + grid, ...: fastest when nothing else runs
    on the GPU) or "dynamic" (workgroups draw tiles from eight XCD queues, so CUs held by another kernel -- an RCCL collective
    overlapping backward -- cost their share, not a second pass: tools/exp/contention.py; +0.5 % without contention, and the
    dynamic instantiation's chunk loop is 2.6 % slower).  PTrainer selects "dynamic" whenever its gradient exchange is active."""
    global _TILE_SCHEDULE
    if mode not in ("dynamic", "static"):
        raise ValueError(f"unknown tile schedule {mode!r}")
    _TILE_SCHEDULE = mode


_WGRAD_WAVES = 1                # fills of the one-workgroup-per-CU slots by the Winograd-domain weight-gradient kernels


_P8_CONV_WAVES = 1               # ... and by the persistent bf16-storage convolution (ptmi_p8_conv3x3_waves)


def set_p8_conv_waves(waves: int) -> None:
    """1 (default): one persistent workgroup per CU; > 1: that many fills, each workgroup walking a 1 / waves share of the tiles --
    PTrainer selects 16 when its gradient exchange is active."""
    global _P8_CONV_WAVES
    if not 1 <= int(waves) <= 64:
        raise ValueError(f"p8 conv waves {waves!r} outside 1 .. 64")
    _P8_CONV_WAVES = int(waves)


def set_wgrad_waves(waves: int) -> None:
    """1 (default): one long workgroup per CU; > 1: that many waves of shorter workgroups (ptmi_conv3x3_wino*_wgrad_waves) -- what
    PTrainer selects when its gradient exchange is active, so that CUs held by the collectives cost their share."""
    global _WGRAD_WAVES
    if not 1 <= int(waves) <= 16:
        raise ValueError(f"wgrad waves {waves!r} outside 1 .. 16")
    _WGRAD_WAVES = int(waves)


def _sched_buf(device):
    """the 16-int32 schedule words of ptmi_conv3x3_wino4_fwd_sched for the CURRENT stream of `device` (zero at creation; the
    kernel leaves them zero) -- one buffer per (device, stream): launches that may overlap must not share one"""
    if _TILE_SCHEDULE != "dynamic":
        return None
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    b = _SCHED_BUFS.get(key)
    if b is None:
        b = _SCHED_BUFS[key] = torch.zeros(16, dtype=torch.int32, device=device)
    return b


def _conv_fwd_call(kind: str, x, wp, bias, mask_ref, y, n, cin, cout, h, w, epilogue):
    if kind == "wino4":
        _lib.call(_CONV_ABI[kind] + "_fwd_sched", _ptr(x), _ptr(wp), _ptr(bias), _ptr(mask_ref), _ptr(y), n, cin, cout, h, w,
                  epilogue, _ptr(_sched_buf(x.device)), _stream())
    else:
        _lib.call(_CONV_ABI[kind] + "_fwd", _ptr(x), _ptr(wp), _ptr(bias), _ptr(mask_ref), _ptr(y), n, cin, cout, h, w, epilogue,
                  _stream())


def conv3x3_raw(x, wp, bias, mask_ref, cout: int, epilogue: int) -> torch.Tensor:
    """wp = conv3x3_pack(w, mode, epilogue) with the SAME epilogue."""
    _no_native_bf16("ops.conv3x3_raw")
    _chk(x, name="conv input")
    n, cin, h, w = x.shape
    y = torch.empty((n, cout, h, w), dtype=F32, device=x.device)
    nbytes = 4.0 * (n * h * w * (cin + cout * (2 if epilogue == 3 else 1)) + 9 * cin * cout)
    kind = _conv_kind(cin, cout, (h, w))
    abi = _CONV_ABI[kind]
    want = getattr(_lib.load(), abi + "_packed_floats")(cin, cout)
    if wp.numel() != want:
        raise _lib.PtmiError(f"conv3x3_raw: packed weights have {wp.numel()} floats, the {kind} kernel this {h}x{w} map is "
                             f"routed to takes {want} (pass hw=(H, W) to conv3x3_pack)")
    with _prof("conv3x3_" + kind, 2.0 * 9 * cin * cout * h * w * n, nbytes, _conv_issued(kind, n, cin, cout, h, w)):
        _conv_fwd_call(kind, x, wp, bias, mask_ref, y, n, cin, cout, h, w, epilogue)
    return y


def conv3x3_relu_pool_nograd(x, weight, bias) -> torch.Tensor:
    """conv3x3 + bias + ReLU + MaxPool(2,2) in ONE kernel (epilogue 4); inference-only (no autograd state)."""
    if _native_bf16():
        from . import p8
        return p8.conv3x3_relu_pool_nograd_nchw(_chk(x.contiguous(), name="conv input"), weight, bias)
    x = _chk(_rnd(x).contiguous(), name="conv input")
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    wp = conv3x3_pack(_chk(_rnd(weight).contiguous()), 0, 4, (h, w))
    y = torch.empty((n, cout, h // 2, w // 2), dtype=F32, device=x.device)
    kind = _conv_kind(cin, cout, (h, w))
    with _prof("conv3x3_" + kind, 2.0 * 9 * cin * cout * h * w * n, 4.0 * (n * h * w * (cin + cout / 4.0) + 9 * cin * cout),
               _conv_issued(kind, n, cin, cout, h, w)):
        _conv_fwd_call(kind, x, wp, _chk(bias.contiguous()), None, y, n, cin, cout, h, w, 4)
    return _rnd_stored(y)


def relu_bwd(dy: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    dz = torch.empty_like(dy)
    if dy.numel() == 0:
        return dz
    with _prof("relu_bwd"):
        _lib.call("ptmi_relu_bwd", _ptr(_chk(dy.contiguous())), _ptr(_chk(y)), _ptr(dz), dy.numel(), _stream())
    return dz


class _Conv3x3(torch.autograd.Function):
    """conv3x3 s1 p1 + bias (+ ReLU).  Replaces Conv2d + F.relu_ (vgg.py:45-53,66-69)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu: bool):
        x = _chk(_rnd(x).contiguous(), name="conv input")
        weight = _chk(_rnd(weight).contiguous(), name="conv weight")
        bias = _chk(bias.contiguous(), name="conv bias")
        wp = conv3x3_pack(weight, 0, 1 if relu else 0, x.shape[-2:])
        y = _rnd_stored(conv3x3_raw(x, wp, bias, None, weight.shape[0], 1 if relu else 0))
        ctx.relu = relu
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = _chk(dy.contiguous(), name="conv grad")
        dz = _rnd(relu_bwd(dy, y) if ctx.relu else dy)
        n, cin, h, w = x.shape
        cout = weight.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw, db = conv3x3_wgrad(x, dz, cout)
        if ctx.needs_input_grad[0]:
            wpd = conv3x3_pack(weight, 1, 2, (h, w))
            dx = _rnd_stored(conv3x3_raw(dz, wpd, None, None, cin, 2))
        return dx, dw, db, None


def conv3x3(x, weight, bias, relu: bool = True):
    if _native_bf16():
        from . import p8
        return p8.conv3x3_nchw(_chk(x.contiguous(), name="conv input"), weight, bias, relu)
    return _Conv3x3.apply(x, weight, bias, relu)


# ============================================================================ max pool
class _MaxPool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x.contiguous(), name="pool input")
        n, c, h, w = x.shape
        y = torch.empty((n, c, h // 2, w // 2), dtype=F32, device=x.device)
        with _prof("maxpool_fwd"):
            _lib.call("ptmi_maxpool2x2_fwd", _ptr(x), _ptr(y), n * c, h, w, _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n, c, h, w = x.shape
        dx = torch.empty_like(x)
        _lib.call("ptmi_maxpool2x2_bwd", _ptr(x), _ptr(_chk(dy.contiguous())), _ptr(dx), n * c, h, w, 0, _stream())
        return dx


def maxpool2x2(x):
    return _MaxPool2x2.apply(x)


class _VGGBlock(torch.autograd.Function):
    """One VGG block = k x [conv3x3 + bias + ReLU] (+ MaxPool 2x2) as a single autograd node (vgg.py:65-72).

    The backward chain is fused: max-pool backward also applies the last ReLU's mask, and every dgrad launch
    applies the mask of the producing layer's ReLU in its epilogue (epilogue 3), so no separate ReLU-backward
    passes (read dy + read y + write dz per layer) remain inside a block."""

    @staticmethod
    def forward(ctx, x, pool: bool, *wb):
        k = len(wb) // 2
        acts = [_chk(x.contiguous(), name="block input")]
        ws = []
        for j in range(k):
            w, b = _chk(wb[2 * j].contiguous()), _chk(wb[2 * j + 1].contiguous())
            ws.append(w)
            acts.append(conv3x3_raw(acts[-1], conv3x3_pack(w, 0, 1, acts[-1].shape[-2:]), b, None, w.shape[0], 1))
        out = acts[-1]
        if pool:
            n, c, h, wd = out.shape
            pooled = torch.empty((n, c, h // 2, wd // 2), dtype=F32, device=out.device)
            with _prof("maxpool_fwd"):
                _lib.call("ptmi_maxpool2x2_fwd", _ptr(out), _ptr(pooled), n * c, h, wd, _stream())
            out = pooled
        ctx.k, ctx.pool = k, pool
        ctx.save_for_backward(*acts, *ws)
        return out

    @staticmethod
    def backward(ctx, dout):
        k = ctx.k
        saved = ctx.saved_tensors
        acts, ws = saved[: k + 1], saved[k + 1:]
        dout = _chk(dout.contiguous())
        yk = acts[k]
        if ctx.pool:
            n, c, h, w = yk.shape
            dz = torch.empty_like(yk)
            _lib.call("ptmi_maxpool2x2_bwd", _ptr(yk), _ptr(dout), _ptr(dz), n * c, h, w, 1, _stream())
        else:
            dz = relu_bwd(dout, yk)
        grads = [None] * (2 * k)
        dx = None
        for j in range(k, 0, -1):
            xin, w = acts[j - 1], ws[j - 1]
            n, cin, h, wd = xin.shape
            cout = w.shape[0]
            if ctx.needs_input_grad[2 + 2 * (j - 1)] or ctx.needs_input_grad[3 + 2 * (j - 1)]:
                dw, db = conv3x3_wgrad(xin, dz, cout)
                grads[2 * (j - 1)], grads[2 * (j - 1) + 1] = dw, db
            if j > 1:
                dz = conv3x3_raw(dz, conv3x3_pack(w, 1, 3, (h, wd)), None, xin, cin, 3)      # dgrad + ReLU mask of layer j-1
            elif ctx.needs_input_grad[0]:
                dx = conv3x3_raw(dz, conv3x3_pack(w, 1, 2, (h, wd)), None, None, cin, 2)
        return (dx, None, *grads)


def vgg_block(x, pool: bool, params):
    """params = [w1, b1, w2, b2, ...]."""
    if _native_bf16():
        from . import p8
        return p8.vgg_block_nchw(_chk(x.contiguous(), name="block input"), pool, params)
    if _OPERAND_ROUNDING == "bf16_emulate":    # layer by layer: every conv rounds its own operands
        for j in range(len(params) // 2):
            x = conv3x3(x, params[2 * j], params[2 * j + 1], True)
        return maxpool2x2(x) if pool else x
    return _VGGBlock.apply(x, pool, *params)


# ============================================================================ GEMM family
def gemm(a, b, m, n, k, lda, ldb, ta, tb, bias=None, bias_mode=0, relu=False, out=None, accumulate=False,
         batch=1, stride_a=0, stride_b=0, stride_c=0, ldc=None):
    """C = op(A) op(B) (+bias)(relu); see ptmi_gemm_f32."""
    if out is None:
        shape = (m, n) if batch == 1 else (batch, m, n)
        out = torch.empty(shape, dtype=F32, device=a.device)
    ldc = n if ldc is None else ldc
    if m == 0 or n == 0:
        return out
    nws = _lib.load().ptmi_gemm_ws_floats(m, n, k, batch)                   # > 0: this shape runs split-K
    ws = _ws("gemm", nws * 4, a.device) if nws else None
    with _prof("gemm_f32", 2.0 * m * n * k * batch):
        _lib.call("ptmi_gemm_bf16" if _native_bf16() else "ptmi_gemm_f32", _ptr(a), _ptr(b), _ptr(out), _ptr(bias), m, n, k, lda, ldb, ldc, ta, tb, bias_mode,
                  int(relu), int(accumulate), batch, stride_a, stride_b, stride_c, _ptr(ws), int(nws), _stream())
    return out


def colsum(a: torch.Tensor) -> torch.Tensor:
    rows, cols = a.shape
    out = torch.empty(cols, dtype=F32, device=a.device)
    if rows == 0:
        return out.zero_()
    nws = _lib.load().ptmi_colsum_ws_floats(rows, cols)                     # > 0: tall matrix, summed by row ranges
    ws = _ws("colsum", nws * 4, a.device) if nws else None
    _lib.call("ptmi_colsum_ws", _ptr(a), _ptr(out), _ptr(ws), rows, cols, 0, _stream())
    return out


class _Linear(torch.autograd.Function):
    """y = x W^T + b (+ReLU).  Replaces nn.Linear/F.relu of FastRCNNConvFCHead + predictors."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu: bool):
        x = _chk(_rnd(x).contiguous(), name="linear input")
        weight = _chk(_rnd(weight).contiguous())
        bias = _chk(bias.contiguous())
        r, k = x.shape
        nout = weight.shape[0]
        y = gemm(x, weight, r, nout, k, k, k, 0, 1, bias=bias, bias_mode=2, relu=relu)
        ctx.relu = relu
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = _chk(dy.contiguous())
        dz = _rnd_grad(relu_bwd(dy, y) if ctx.relu else dy)
        r, k = x.shape
        nout = weight.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm(dz, weight, r, k, nout, nout, k, 0, 0)                  # (r,nout) x (nout,k)
        if ctx.needs_input_grad[1]:
            if r == 0:
                dw = torch.zeros_like(weight)
            else:
                dw = gemm(dz, x, nout, k, r, nout, k, 1, 0)                   # dz^T (nout,r) x x (r,k)
        if ctx.needs_input_grad[2]:
            db = colsum(dz)
        return dx, dw, db, None


def linear(x, weight, bias, relu: bool = False):
    if _native_bf16() and x.shape[0] > 0:
        from . import p8
        if weight.shape[1] >= p8.LINEAR_MIN_K:          # fc1 (25088 -> 1024): bf16-storage GEMMs
            return p8.linear(x, weight, bias, relu)
    return _Linear.apply(x, weight, bias, relu)


class _Conv1x1(torch.autograd.Function):
    """1x1 conv as a batched GEMM over images (D2 StandardRPNHead objectness / anchor-delta convs)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _chk(_rnd(x).contiguous())
        n, ci, h, w = x.shape
        co = weight.shape[0]
        w2 = _chk(_rnd(weight).reshape(co, ci).contiguous())
        hw = h * w
        y = torch.empty((n, co, h, w), dtype=F32, device=x.device)
        gemm(w2, x, co, hw, ci, ci, hw, 0, 0, bias=_chk(bias.contiguous()), bias_mode=1, out=y, batch=n, stride_a=0,
             stride_b=ci * hw, stride_c=co * hw, ldc=hw)
        ctx.save_for_backward(x, w2)
        ctx.wshape = weight.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        dy = _chk(_rnd_grad(dy).contiguous())
        n, ci, h, w = x.shape
        co, hw = w2.shape[0], h * w
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm(w2, dy, ci, hw, co, ci, hw, 1, 0, out=dx, batch=n, stride_a=0, stride_b=co * hw, stride_c=ci * hw,
                 ldc=hw)
        if ctx.needs_input_grad[1]:
            # per-image partials in ONE batched launch, then a fixed-order sum over images
            part = gemm(dy, x, co, ci, hw, hw, hw, 0, 1, batch=n, stride_a=co * hw, stride_b=ci * hw,
                        stride_c=co * ci)
            dw = colsum(part.view(n, co * ci)).view(ctx.wshape)
        if ctx.needs_input_grad[2]:
            db = torch.empty(co, dtype=F32, device=x.device)
            _lib.call("ptmi_rowsum_batched", _ptr(dy), _ptr(db), n, co, hw, 0, _stream())
        return dx, dw, db


class _RPNHead1x1(torch.autograd.Function):
    """The two 1x1 convolutions of the RPN head (D2 StandardRPNHead: objectness 512 -> A, anchor deltas 512 -> 8A) WRITING
    THE LAYOUTS THE RPN CONSUMES: the reference permutes (N, A, H, W) -> (N, HWA) and (N, 8A, H, W) -> (N, HWA, 8)
    (rpn.py:97-113) with anchor index (y w + x) A + a -- i.e. the transposed GEMM.  Here each image's output is computed as
    X^T (HW x 512) . W^T (512 x A | 8A) with the bias per column, so the GEMM's row-major C IS that layout: no permute
    copies in forward, none of their transposes in backward."""

    @staticmethod
    def forward(ctx, x, w_obj, b_obj, w_del, b_del):
        x = _chk(_rnd(x).contiguous())
        n, ci, h, w = x.shape
        hw = h * w
        outs, w2s = [], []
        for wt, b in ((w_obj, b_obj), (w_del, b_del)):
            co = wt.shape[0]
            w2 = _chk(_rnd(wt).reshape(co, ci).contiguous())
            y = torch.empty((n, hw, co), dtype=F32, device=x.device)
            # A = x[n] stored (K = ci, M = hw): ta = 1;  B = W stored (N = co, K = ci): tb = 1;  C (hw, co)
            gemm(x, w2, hw, co, ci, hw, ci, 1, 1, bias=_chk(b.contiguous()), bias_mode=2, out=y, batch=n, stride_a=ci * hw,
                 stride_b=0, stride_c=hw * co, ldc=co)
            outs.append(y)
            w2s.append(w2)
        ctx.save_for_backward(x, *w2s)
        ctx.shapes = (w_obj.shape, w_del.shape)
        a = w_obj.shape[0]
        return outs[0].view(n, hw * a), outs[1].view(n, hw * a, 8)

    @staticmethod
    def backward(ctx, g_obj, g_del):
        x, w_o, w_d = ctx.saved_tensors
        n, ci, h, w = x.shape
        hw = h * w
        dx = None
        grads = []
        first = True
        for g, w2, shape in ((g_obj, w_o, ctx.shapes[0]), (g_del, w_d, ctx.shapes[1])):
            co = w2.shape[0]
            dy = _chk(_rnd_grad(g.reshape(n, hw, co)).contiguous())
            if ctx.needs_input_grad[0]:
                if dx is None:
                    dx = torch.empty_like(x)
                # dX (ci, hw) = W^T (ci, co) . dY^T (co, hw):  A = W stored (K = co, M = ci): ta = 1;  B = dY stored (N = hw, K = co): tb = 1
                gemm(w2, dy, ci, hw, co, ci, co, 1, 1, out=dx, accumulate=not first, batch=n, stride_a=0, stride_b=hw * co,
                     stride_c=ci * hw, ldc=hw)
                first = False
            # dW (co, ci) = dY^T (co, hw) . X^T (hw, ci) per image, then a fixed-order sum over the images
            part = gemm(dy, x, co, ci, hw, co, hw, 1, 1, batch=n, stride_a=hw * co, stride_b=ci * hw, stride_c=co * ci)
            grads.append(colsum(part.view(n, co * ci)).view(shape))
            grads.append(colsum(dy.view(n * hw, co)))
        return dx, grads[0], grads[1], grads[2], grads[3]


def rpn_head_1x1(x, w_obj, b_obj, w_del, b_del):
    """-> (objectness logits (N, H W A), anchor deltas (N, H W A, 8)) in the RPN's anchor order"""
    return _RPNHead1x1.apply(x, w_obj, b_obj, w_del, b_del)


def conv1x1(x, weight, bias):
    return _Conv1x1.apply(x, weight, bias)


# ============================================================================ ROIAlign
class _ROIAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, pooled: int, scale: float, img_offsets):
        feat = _chk(feat.contiguous())
        rois = _chk(rois.contiguous())
        n, c, h, w = feat.shape
        r = rois.shape[0]
        out = torch.empty((r, c, pooled, pooled), dtype=F32, device=feat.device)
        with _prof("roi_align_fwd"):
            if img_offsets is not None:
                ws = torch.empty(_lib.load().ptmi_roi_align_ws_bytes(r, h, w), dtype=torch.uint8, device=feat.device)
                _lib.call("ptmi_roi_align_fwd_grouped", _ptr(feat), _ptr(rois), _ptr(_chk(img_offsets, torch.int32)),
                          _ptr(out), _ptr(ws), n, c, h, w, r, pooled, float(scale), _stream())
            else:
                _lib.call("ptmi_roi_align_fwd", _ptr(feat), _ptr(rois), _ptr(out), n, c, h, w, r, pooled, float(scale),
                          _stream())
        ctx.save_for_backward(rois, img_offsets)
        ctx.meta = (n, c, h, w, pooled, float(scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        rois, img_offsets = ctx.saved_tensors
        n, c, h, w, pooled, scale = ctx.meta
        dout = _chk(dout.contiguous())
        if img_offsets is not None:
            # rois grouped by image: LDS-accumulating kernel, no global atomics, writes every element of dfeat
            dfeat = roi_align_bwd_grouped(dout, rois, img_offsets, n, c, h, w, pooled, scale)
        else:
            dfeat = torch.zeros((n, c, h, w), dtype=F32, device=dout.device)
            with _prof("roi_align_bwd"):
                _lib.call("ptmi_roi_align_bwd", _ptr(dout), _ptr(rois), _ptr(dfeat), n, c, h, w, rois.shape[0], pooled,
                          scale, _stream())
        return dfeat, None, None, None, None


def roi_align_bwd_grouped(dout: torch.Tensor, rois: torch.Tensor, img_offsets: torch.Tensor, n: int, c: int, h: int, w: int,
                          pooled: int, scale: float) -> torch.Tensor:
    """d(feature map) (n, c, h, w) from the gradient of ROIAlign's output (R, c * pooled * pooled) for rois grouped by image"""
    dout = _chk(dout.contiguous())
    dfeat = torch.empty((n, c, h, w), dtype=F32, device=dout.device)
    with _prof("roi_align_bwd"):
        ws = torch.empty(_lib.load().ptmi_roi_align_bwd_ws_bytes(rois.shape[0], h, w), dtype=torch.uint8, device=dout.device)
        _lib.call("ptmi_roi_align_bwd_grouped", _ptr(dout), _ptr(rois), _ptr(_chk(img_offsets, torch.int32)), _ptr(dfeat), _ptr(ws),
                  n, c, h, w, rois.shape[0], pooled, scale, _stream())
    return dfeat


def roi_align_p8m_fits(c: int, h: int, w: int, pooled: int) -> bool:
    return bool(_lib.load().ptmi_roi_align_fwd_p8m_fits(int(c), int(h), int(w), int(pooled)))


def roi_align_p8m(feat: torch.Tensor, rois: torch.Tensor, img_offsets: torch.Tensor, pooled: int, scale: float, need_xt: bool):
    """ROIAlign (rois grouped by image) straight into the bf16 "P8 matrix" operands of the box head's first Linear layer:
    xk (c * pooled^2 / 8, R, 8) and, if need_xt, xt (ceil(R / 8), c * pooled^2, 8)  (ptmi_roi_align_fwd_p8m)"""
    feat = _chk(feat.contiguous())
    rois = _chk(rois.contiguous())
    n, c, h, w = feat.shape
    r, kd = rois.shape[0], c * pooled * pooled
    xk = torch.empty((kd // 8, r, 8), dtype=torch.bfloat16, device=feat.device)
    xt = torch.empty((-(-r // 8), kd, 8), dtype=torch.bfloat16, device=feat.device) if need_xt else None
    if r:
        with _prof("roi_align_fwd"):
            ws = torch.empty(_lib.load().ptmi_roi_align_ws_bytes(r, h, w), dtype=torch.uint8, device=feat.device)
            _lib.call("ptmi_roi_align_fwd_p8m", _ptr(feat), _ptr(rois), _ptr(_chk(img_offsets, torch.int32)), _ptr(xk), _ptr(xt), _ptr(ws),
                      n, c, h, w, r, pooled, float(scale), _stream())
    return xk, xt


def roi_align(feat, rois, pooled: int, scale: float, img_offsets=None):
    """rois (R,5) = [image index, x1, y1, x2, y2].  If the rows are grouped by image, pass `img_offsets`
    (int32 (N+1,) device tensor of row offsets) to enable the atomic-free backward."""
    return _ROIAlign.apply(feat, rois, pooled, scale, img_offsets)


# ============================================================================ boxes
def grid_anchors(cell: torch.Tensor, h: int, w: int, stride: float, offset: float) -> torch.Tensor:
    cell = _chk(cell.contiguous())
    a = cell.shape[0]
    out = torch.empty((h * w * a, 4), dtype=F32, device=cell.device)
    _lib.call("ptmi_grid_anchors", _ptr(cell), _ptr(out), h, w, a, float(stride), float(offset), _stream())
    return out


def apply_deltas(deltas: torch.Tensor, boxes: torch.Tensor, weights: Sequence[float], scale_clamp: float,
                 k: Optional[int] = None, dstride: Optional[int] = None) -> torch.Tensor:
    """deltas (rows, >=4k) [row stride dstride], boxes (nb,4) cycled over rows -> (rows, 4k)."""
    _chk(deltas)
    boxes = _chk(boxes.contiguous())
    rows = deltas.shape[0]
    dstride = deltas.shape[1] if dstride is None else dstride
    k = deltas.shape[1] // 4 if k is None else k
    out = torch.empty((rows, 4 * k), dtype=F32, device=deltas.device)
    wx, wy, ww, wh = [float(v) for v in weights]
    if rows:
        _lib.call("ptmi_apply_deltas", _ptr(deltas), _ptr(boxes), _ptr(out), rows, k, dstride, boxes.shape[0], wx, wy,
                  ww, wh, float(scale_clamp), _stream())
    return out


class _GetDeltas(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, tgt, wx, wy, ww, wh):
        src = _chk(src.contiguous())
        tgt = _chk(tgt.contiguous())
        out = torch.empty_like(src)
        if src.shape[0]:
            _lib.call("ptmi_get_deltas", _ptr(src), _ptr(tgt), _ptr(out), src.shape[0], wx, wy, ww, wh, _stream())
        ctx.save_for_backward(src, tgt)
        ctx.w = (wx, wy, ww, wh)
        return out

    @staticmethod
    def backward(ctx, dd):
        src, tgt = ctx.saved_tensors
        dsrc = None
        if ctx.needs_input_grad[0]:
            rows = src.shape[0]
            dsrc = torch.zeros_like(src)
            if rows:
                idx = torch.arange(rows, dtype=torch.int64, device=src.device)
                _lib.call("ptmi_get_deltas_bwd_src", _ptr(src), _ptr(tgt), _ptr(_chk(dd.contiguous())), _ptr(idx), rows,
                          *ctx.w, _ptr(dsrc), _stream())
        return dsrc, None, None, None, None, None


def get_deltas(src, tgt, weights: Sequence[float]):
    wx, wy, ww, wh = [float(v) for v in weights]
    return _GetDeltas.apply(src, tgt, wx, wy, ww, wh)


def iou_match(gt: torch.Tensor, boxes: torch.Tensor, thresholds: Sequence[float], labels: Sequence[int],
              allow_low_quality: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """pairwise_iou + Matcher fused -> (matched_idx int64, matched_label int8, matched_iou f32)."""
    boxes = _chk(boxes.contiguous())
    gt = _chk(gt.contiguous())
    nb, m = boxes.shape[0], gt.shape[0]
    dev = boxes.device
    midx = torch.empty(nb, dtype=torch.int64, device=dev)
    mlab = torch.empty(nb, dtype=torch.int8, device=dev)
    miou = torch.empty(nb, dtype=F32, device=dev)
    ws = _ws("ioum", max(m, 1) * 4, dev)
    thr = (ctypes.c_float * len(thresholds))(*[float(t) for t in thresholds])
    lab = (ctypes.c_int * len(labels))(*[int(l) for l in labels])
    _lib.call("ptmi_iou_match", _ptr(gt), _ptr(boxes), m, nb, thr, lab, len(thresholds), int(allow_low_quality),
              _ptr(midx), _ptr(mlab), _ptr(miou), _ptr(ws), _stream())
    return midx, mlab, miou


def dev_i32(values, device) -> torch.Tensor:
    """Small host list -> int32 device tensor without a stream synchronisation (pinned staging, async copy)."""
    return torch.tensor(values, dtype=torch.int32).pin_memory().to(device, non_blocking=True)


def iou_match_batched(gt_all: torch.Tensor, gt_counts: Sequence[int], boxes: torch.Tensor, box_counts, thresholds,
                      labels, allow_low_quality: bool):
    """ptmi_iou_match for a whole batch in one launch per pass.  gt_all: the images' gt boxes concatenated
    (gt_counts[i] rows each).  boxes: concatenated per image (box_counts[i] rows each) or, with box_counts=None, ONE
    box set shared by all images (outputs (N, nb)).  Returns (matched_idx within the image, label int8, iou)."""
    boxes = _chk(boxes.contiguous())
    gt_all = _chk(gt_all.contiguous())
    dev = boxes.device
    n = len(gt_counts)
    goff = [0]
    for c in gt_counts:
        goff.append(goff[-1] + int(c))
    gt_off = dev_i32(goff, dev)
    if box_counts is None:
        nb, box_off, shape = boxes.shape[0], None, (n, boxes.shape[0])
    else:
        boff = [0]
        for c in box_counts:
            boff.append(boff[-1] + int(c))
        nb, box_off, shape = max(box_counts) if len(box_counts) else 0, dev_i32(boff, dev), (boff[-1],)
    midx = torch.empty(shape, dtype=torch.int64, device=dev)
    mlab = torch.empty(shape, dtype=torch.int8, device=dev)
    miou = torch.empty(shape, dtype=F32, device=dev)
    ws = _ws("ioumb", max(goff[-1], 1) * 4, dev)
    thr = (ctypes.c_float * len(thresholds))(*[float(t) for t in thresholds])
    lab = (ctypes.c_int * len(labels))(*[int(l) for l in labels])
    if midx.numel():
        _lib.call("ptmi_iou_match_batched", _ptr(gt_all), _ptr(gt_off), _ptr(boxes), _ptr(box_off), n, nb, goff[-1], thr,
                  lab, len(thresholds), int(allow_low_quality), _ptr(midx), _ptr(mlab), _ptr(miou), _ptr(ws), _stream())
    return midx, mlab, miou, gt_off, box_off


def sample_by_keys(cls_all: torch.Tensor, keys_all: torch.Tensor, offsets: torch.Tensor, max_count: int,
                   num_samples: int, num_pos_max: int, bg_label: int):
    """D2 subsample_labels for a batch without host syncs (ptmi_sample_by_keys): (fg (N,num_pos_max) int64, bg
    (N,num_samples) int64, counts (N,2) int32); entries beyond the counts are unspecified."""
    cls_all = _chk(cls_all.contiguous(), torch.int64)
    keys_all = _chk(keys_all.contiguous())
    n = offsets.numel() - 1
    dev = cls_all.device
    fg = torch.empty((n, max(num_pos_max, 1)), dtype=torch.int64, device=dev)
    bg = torch.empty((n, num_samples), dtype=torch.int64, device=dev)
    cnt = torch.empty((n, 2), dtype=torch.int32, device=dev)
    _lib.call("ptmi_sample_by_keys", _ptr(cls_all), _ptr(keys_all), _ptr(_chk(offsets, torch.int32)), n, int(max_count),
              num_samples, num_pos_max, bg_label, _ptr(fg), _ptr(bg), _ptr(cnt), _stream())
    return fg, bg, cnt


def rpn_subsample_relabel(labels: torch.Tensor, keys: torch.Tensor, num_samples: int, num_pos_max: int,
                          bg_label: int) -> torch.Tensor:
    """RPN._subsample_labels for a batch (ptmi_rpn_subsample_relabel): labels (N, R) int8, keys (N, R) float32 >= 0 ->
    (N, R) int8 with the sampled positives = 1, sampled negatives = 0, everything else -1."""
    labels = _chk(labels.contiguous(), torch.int8)
    keys = _chk(keys.contiguous())
    assert labels.dim() == 2 and keys.shape == labels.shape
    out = torch.empty_like(labels)
    if labels.numel():
        _lib.call("ptmi_rpn_subsample_relabel", _ptr(labels), _ptr(keys), _ptr(out), labels.shape[0], labels.shape[1],
                  int(num_samples), int(num_pos_max), int(bg_label), _stream())
    return out


# ============================================================================ sort / proposals / NMS
def segsort_desc(keys: torch.Tensor, seg_offsets: torch.Tensor, max_len: Optional[int] = None,
                 topk: Optional[int] = None, lengths: Optional[Sequence[int]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Stable descending sort inside each segment.  Returns (sorted keys, index within segment int32).
    max_len: the longest segment, if the caller knows it (no device read here) -- segments of up to 16 384 keys are then sorted
    by the LDS kernel; topk: the caller only reads the first topk entries of every segment (longer segments come back with
    their first topk entries in order and (-inf, 0) behind them).  lengths: the HOST copy of the segment lengths the caller
    built seg_offsets from -- checked here against max_len and the key count (a max_len that does not describe the
    segments makes the LDS kernel emit NaN keys, which only one of the three callers would have noticed; all three now pass
    their host lengths, so the mismatch is a PtmiError before any launch)."""
    keys = _chk(keys.contiguous())
    seg_offsets = _chk(seg_offsets.contiguous(), torch.int32)
    total, nseg = keys.numel(), seg_offsets.numel() - 1
    if lengths is not None:
        longest = max(lengths) if len(lengths) else 0
        if len(lengths) != nseg or sum(lengths) != total or (max_len is not None and longest > max_len):
            raise _lib.PtmiError(f"segsort_desc: segment lengths {list(lengths)[:8]}... (n = {len(lengths)}, sum {sum(lengths)}, longest "
                                 f"{longest}) do not match {nseg} segments / {total} keys / max_len {max_len}")
        max_len = longest if max_len is None else max_len
    out = torch.empty_like(keys)
    idx = torch.empty(total, dtype=torch.int32, device=keys.device)
    if total == 0:
        return out, idx
    lib = _lib.load()
    if max_len is not None and lib.ptmi_segsort_topk_fits(int(max_len), int(topk or 0)):
        with _prof("segsort_desc"):
            _lib.call("ptmi_segsort_topk_desc", _ptr(keys), _ptr(out), _ptr(idx), nseg, _ptr(seg_offsets), int(max_len),
                      int(topk or 0), _stream())
        return out, idx
    nbytes = lib.ptmi_segsort_ws_bytes(total, nseg)
    ws = _ws("sort", nbytes, keys.device)
    with _prof("segsort_desc"):
        _lib.call("ptmi_segsort_desc", _ptr(keys), _ptr(out), _ptr(idx), total, nseg, _ptr(seg_offsets), _ptr(ws),
                  nbytes, _stream())
    return out, idx


def rpn_prepare(decoded, sorted_logits, sorted_idx, sigma_logits, image_sizes_hw, k: int, min_size: float):
    """-> boxes (n,k,4) clipped, keys (n,k) (rescored logit, -inf for dropped entries), counts (n,) int32, nonfinite (n,)"""
    n, r = sorted_logits.shape
    dev = decoded.device
    boxes = torch.empty((n, k, 4), dtype=F32, device=dev)
    keys = torch.empty((n, k), dtype=F32, device=dev)
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    nonfinite = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.call("ptmi_rpn_prepare", _ptr(_chk(decoded)), _ptr(_chk(sorted_logits)), _ptr(_chk(sorted_idx, torch.int32)),
              _ptr(_chk(sigma_logits)), _ptr(_chk(image_sizes_hw)), _ptr(boxes), _ptr(keys), _ptr(counts),
              _ptr(nonfinite), n, r, k, float(min_size), _stream())
    return boxes, keys, counts, nonfinite


def nms_batched(boxes_sorted: torch.Tensor, seg_offsets: torch.Tensor, max_count: int, thr: float, max_keep: int,
                seg_counts: Optional[torch.Tensor] = None):
    """boxes (sum,4) sorted by descending score per image; returns keep (nimg,max_keep) int32 positions, counts.
    seg_counts (nimg,) int32: fill level of fixed-capacity segments (device-side, no host read needed)."""
    boxes_sorted = _chk(boxes_sorted.contiguous())
    seg_offsets = _chk(seg_offsets.contiguous(), torch.int32)
    nimg = seg_offsets.numel() - 1
    dev = boxes_sorted.device
    keep = torch.empty((nimg, max_keep), dtype=torch.int32, device=dev)
    cnt = torch.empty(nimg, dtype=torch.int32, device=dev)
    nbytes = _lib.load().ptmi_nms_ws_bytes(max_count, nimg)
    ws = _ws("nms", nbytes, dev)
    with _prof("nms_batched"):
        _lib.call("ptmi_nms_batched", _ptr(boxes_sorted), _ptr(seg_offsets),
                  _ptr(_chk(seg_counts, torch.int32)) if seg_counts is not None else None, nimg, int(max_count),
                  float(thr), max_keep, _ptr(keep), _ptr(cnt), _ptr(ws), _stream())
    return keep, cnt


def roi_infer_prepare(deltas, proposal_boxes, probs, roi_img, image_sizes_hw, k: int, weights, scale_clamp: float,
                      score_thresh: float):
    """fast_rcnn.py:34-101 for all ROIs of a batch (ptmi_roi_infer_prepare): -> boxes (R,K,4) clipped, keys (R,K),
    roi_valid (R,) u8, img_max (nimg,), img_count (nimg,) int32, img_invalid (nimg,) int32."""
    r, nimg = deltas.shape[0], image_sizes_hw.shape[0]
    dev = deltas.device
    boxes = torch.empty((r, k, 4), dtype=F32, device=dev)
    keys = torch.empty((r, k), dtype=F32, device=dev)
    valid = torch.empty(r, dtype=torch.uint8, device=dev)
    img_max = torch.empty(nimg, dtype=F32, device=dev)
    img_cnt = torch.empty(nimg, dtype=torch.int32, device=dev)
    img_inv = torch.empty(nimg, dtype=torch.int32, device=dev)
    wx, wy, ww, wh = [float(v) for v in weights]
    _lib.call("ptmi_roi_infer_prepare", _ptr(_chk(deltas)), _ptr(_chk(proposal_boxes)), _ptr(_chk(probs)),
              _ptr(_chk(roi_img, torch.int32)), _ptr(_chk(image_sizes_hw)), _ptr(boxes), _ptr(keys), _ptr(valid),
              _ptr(img_max), _ptr(img_cnt), _ptr(img_inv), r, k, nimg, wx, wy, ww, wh, float(scale_clamp),
              float(score_thresh), _stream())
    return boxes, keys, valid, img_max, img_cnt, img_inv


def roi_infer_nms_boxes(boxes, order, seg_offsets, img_max, max_count: int, k: int) -> torch.Tensor:
    out = torch.empty((boxes.shape[0] * boxes.shape[1], 4), dtype=F32, device=boxes.device)
    _lib.call("ptmi_roi_infer_nms_boxes", _ptr(_chk(boxes)), _ptr(_chk(order, torch.int32)),
              _ptr(_chk(seg_offsets, torch.int32)), _ptr(_chk(img_max)), seg_offsets.numel() - 1, int(max_count), k,
              _ptr(out), _stream())
    return out


# ============================================================================ losses (loss + gradient in one launch)
class _BCELogitsSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, inv_norm: float):
        logits = _chk(logits.contiguous())
        labels = _chk(labels.contiguous(), torch.int8)
        loss = torch.empty(1, dtype=F32, device=logits.device)
        dl = torch.empty_like(logits)
        _lib.call("ptmi_bce_logits_sum", _ptr(logits), _ptr(labels), logits.numel(), float(inv_norm), _ptr(loss),
                  _ptr(dl), _ptr(_loss_ws(logits.device)), _stream())
        ctx.save_for_backward(dl)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None


def bce_logits_sum(logits, labels_i8, inv_norm):
    return _BCELogitsSum.apply(logits, labels_i8, inv_norm)


class _GaussianNLLSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, t, inv_norm: float):
        d = _chk(d.contiguous())
        t = _chk(t.contiguous())
        rows = d.shape[0]
        loss = torch.empty(1, dtype=F32, device=d.device)
        dd = torch.empty_like(d)
        dt = torch.empty_like(t) if ctx.needs_input_grad[1] else None
        _lib.call("ptmi_gaussian_nll_sum", _ptr(d), _ptr(t), rows, float(inv_norm), _ptr(loss), _ptr(dd), _ptr(dt),
                  _ptr(_loss_ws(d.device)), _stream())
        ctx.has_dt = dt is not None
        ctx.save_for_backward(dd, dt) if dt is not None else ctx.save_for_backward(dd)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if ctx.has_dt:
            dd, dt = ctx.saved_tensors
            return dd * g, dt * g, None
        (dd,) = ctx.saved_tensors
        return dd * g, None, None


def gaussian_nll_sum(d, t, inv_norm):
    return _GaussianNLLSum.apply(d, t, inv_norm)


class _SoftmaxCEMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        logits = _chk(logits.contiguous())
        target = _chk(target.contiguous(), torch.int64)
        r, c = logits.shape
        loss = torch.empty(1, dtype=F32, device=logits.device)
        dl = torch.empty_like(logits)
        _lib.call("ptmi_softmax_ce_mean", _ptr(logits), _ptr(target), r, c, _ptr(loss), _ptr(dl),
                  _ptr(_loss_ws(logits.device)), _stream())
        ctx.save_for_backward(dl)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


def softmax_ce_mean(logits, target):
    return _SoftmaxCEMean.apply(logits, target)


def softmax_rows(logits: torch.Tensor) -> torch.Tensor:
    logits = _chk(logits.contiguous())
    out = torch.empty_like(logits)
    _lib.call("ptmi_softmax_rows", _ptr(logits), _ptr(out), logits.shape[0], logits.shape[1], _stream())
    return out


class _SoftCEEFL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, teacher, student, tau, lam, efl, inv_norm):
        teacher = _chk(teacher.contiguous())
        student = _chk(student.contiguous())
        r, c = student.shape
        loss = torch.empty(1, dtype=F32, device=student.device)
        ds = torch.empty_like(student)
        _lib.call("ptmi_soft_ce_efl", _ptr(teacher), _ptr(student), r, c, float(tau), float(lam), int(efl),
                  float(inv_norm), _ptr(loss), _ptr(ds), _ptr(_loss_ws(student.device)), _stream())
        ctx.save_for_backward(ds)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (ds,) = ctx.saved_tensors
        return None, ds * g, None, None, None, None


def soft_ce_efl(teacher, student, tau, lam, efl, inv_norm):
    return _SoftCEEFL.apply(teacher, student, tau, lam, efl, inv_norm)


class _RPNSoftObj(torch.autograd.Function):
    @staticmethod
    def forward(ctx, teacher, x, tau, lam, efl, inv_norm):
        teacher = _chk(teacher.contiguous())
        x = _chk(x.contiguous())
        k, c = teacher.shape
        loss = torch.empty(1, dtype=F32, device=x.device)
        dx = torch.empty_like(x)
        fg = torch.empty(k, dtype=torch.uint8, device=x.device)
        _lib.call("ptmi_rpn_soft_obj_loss", _ptr(teacher), _ptr(x), k, c, float(tau), float(lam), int(efl),
                  float(inv_norm), _ptr(loss), _ptr(dx), _ptr(fg), _ptr(_loss_ws(x.device)), _stream())
        ctx.save_for_backward(dx)
        ctx.mark_non_differentiable(fg)
        return loss.reshape(()), fg

    @staticmethod
    def backward(ctx, g, _gfg):
        (dx,) = ctx.saved_tensors
        return None, dx * g, None, None, None, None


def rpn_soft_obj_loss(teacher, x, tau, lam, efl, inv_norm):
    return _RPNSoftObj.apply(teacher, x, tau, lam, efl, inv_norm)


class _KLEFL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, mu_p, slog_p, fg, tau, lam, efl, reduction, inv_norm):
        q = _chk(q.contiguous())
        mu_p = _chk(mu_p.contiguous())
        slog_p = _chk(slog_p.contiguous())
        if fg is not None:
            fg = _chk(fg.contiguous(), torch.uint8)
        rows = q.shape[0]
        loss = torch.empty(1, dtype=F32, device=q.device)
        dq = torch.empty_like(q)
        dmu = torch.empty_like(mu_p) if ctx.needs_input_grad[1] else None
        _lib.call("ptmi_kl_efl_loss", _ptr(q), _ptr(mu_p), _ptr(slog_p), _ptr(fg), rows, float(tau), float(lam),
                  int(efl), int(reduction), float(inv_norm), _ptr(loss), _ptr(dq), _ptr(dmu),
                  _ptr(_loss_ws(q.device)), _stream())
        ctx.has_dmu = dmu is not None
        ctx.save_for_backward(dq, dmu) if dmu is not None else ctx.save_for_backward(dq)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if ctx.has_dmu:
            dq, dmu = ctx.saved_tensors
            return dq * g, dmu * g, None, None, None, None, None, None, None
        (dq,) = ctx.saved_tensors
        return dq * g, None, None, None, None, None, None, None, None


def kl_efl_loss(q, mu_p, slog_p, fg, tau, lam, efl, reduction, inv_norm):
    return _KLEFL.apply(q, mu_p, slog_p, fg, tau, lam, efl, reduction, inv_norm)


# ============================================================================ optimiser / EMA / image prep
def ema_update(student_flat: torch.Tensor, teacher_flat: torch.Tensor, keep_rate: float) -> None:
    _lib.call("ptmi_ema_update", _ptr(_chk(student_flat)), _ptr(_chk(teacher_flat)), student_flat.numel(),
              float(keep_rate), float(1 - keep_rate), _stream())


def sumsq(g_flat: torch.Tensor) -> torch.Tensor:
    out = torch.empty(1, dtype=F32, device=g_flat.device)
    ws = _ws("sumsq", 1024 * 4, g_flat.device)
    _lib.call("ptmi_sumsq", _ptr(_chk(g_flat)), g_flat.numel(), _ptr(out), _ptr(ws), _stream())
    return out


def clip_sgd_step(p, g, buf, sumsq_t, clip_norm, lr, momentum, weight_decay, first: bool) -> None:
    _lib.call("ptmi_clip_sgd_step", _ptr(_chk(p)), _ptr(_chk(g)), _ptr(_chk(buf)), p.numel(), _ptr(sumsq_t),
              float(clip_norm), float(lr), float(momentum), float(weight_decay), int(first), _stream())


def scale_by_clip(g, sumsq_t, clip_norm) -> None:
    _lib.call("ptmi_scale_by_clip", _ptr(_chk(g)), g.numel(), _ptr(sumsq_t), float(clip_norm), _stream())


def _image_desc(rows, device) -> torch.Tensor:
    """rows of 8 ints -> device int64 descriptor table (pinned staging, async copy: no stream synchronisation)."""
    return torch.tensor(rows, dtype=torch.int64).pin_memory().to(device, non_blocking=True)


def preprocess_images(images_u8: List[torch.Tensor], mean: Sequence[float], std: Sequence[float]):
    """D2 preprocess_image + ImageList.from_tensors: (x-mean)/std, zero pad to the batch max; ONE launch per batch."""
    hmax = max(im.shape[-2] for im in images_u8)
    wmax = max(im.shape[-1] for im in images_u8)
    dev = images_u8[0].device
    images_u8 = [_chk(im.contiguous(), torch.uint8, "image") for im in images_u8]
    out = torch.empty((len(images_u8), 3, hmax, wmax), dtype=F32, device=dev)
    desc = _image_desc([[im.data_ptr(), 0, im.shape[-2], im.shape[-1], 0, 0, 0, 0] for im in images_u8], dev)
    _lib.call("ptmi_preprocess_batched", _ptr(desc), _ptr(out), len(images_u8), hmax, wmax, float(mean[0]), float(mean[1]),
              float(mean[2]), float(std[0]), float(std[1]), float(std[2]), _stream())
    return out


def shrink_paste_geometry(h: int, w: int, ratio: float) -> Tuple[int, int, int, int]:
    """(dh, dw, x1, y1) of trainer.py:563-566."""
    dh, dw = int(h * ratio), int(w * ratio)
    return dh, dw, int((w - dw) / 2), int((h - dh) / 2)


def shrink_paste(img_u8: torch.Tensor, ratio: float, mean_int: Sequence[int]) -> Tuple[torch.Tensor, int, int]:
    """trainer.py:557-590 image part for one image; returns (canvas, x1, y1)."""
    img_u8 = _chk(img_u8.contiguous(), torch.uint8, "image")
    h, w = img_u8.shape[-2:]
    dh, dw, x1, y1 = shrink_paste_geometry(h, w, ratio)
    out = torch.empty_like(img_u8)
    _lib.call("ptmi_shrink_paste", _ptr(img_u8), _ptr(out), h, w, dh, dw, y1, x1, int(mean_int[0]), int(mean_int[1]),
              int(mean_int[2]), _stream())
    return out, x1, y1


def shrink_paste_batch(images_u8: List[torch.Tensor], ratios: Sequence[float], mean_int: Sequence[int]):
    """trainer.py:557-590 image part for a whole batch in ONE launch; returns (canvases, [(x1, y1), ...])."""
    images_u8 = [_chk(im.contiguous(), torch.uint8, "image") for im in images_u8]
    if not images_u8:
        return [], []
    dev = images_u8[0].device
    sizes = [3 * im.shape[-2] * im.shape[-1] for im in images_u8]
    buf = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
    outs, rows, offs, o = [], [], [], 0
    for im, ratio, sz in zip(images_u8, ratios, sizes):
        h, w = im.shape[-2:]
        dh, dw, x1, y1 = shrink_paste_geometry(h, w, ratio)
        canvas = buf[o:o + sz].view(3, h, w)
        o += sz
        outs.append(canvas)
        offs.append((x1, y1))
        rows.append([im.data_ptr(), canvas.data_ptr(), h, w, dh, dw, y1, x1])
    desc = _image_desc(rows, dev)
    _lib.call("ptmi_shrink_paste_batched", _ptr(desc), len(images_u8), max(sizes), int(mean_int[0]), int(mean_int[1]),
              int(mean_int[2]), _stream())
    return outs, offs
