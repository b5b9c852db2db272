"""oracle/augment.py -- TEST INFRASTRUCTURE: the strong augmentation of the reference's mapper on the CPU.

Two implementations of the same pixel arithmetic:
  * `pil_*`   : the literal calls the reference / torchvision 0.8.2 make on a PIL image (needs Pillow; used to pin);
  * `c_*`     : oracle/csrc/ref_aug.c, the restatement the HIP kernels are compared with (no Pillow needed).
Reference: pt/data/detection_utils.py:38-60, pt/data/transforms/augmentation_impl.py, pt/data/dataset_mapper.py:151-159;
torchvision.transforms.functional_pil adjust_brightness/contrast/saturation/hue, to_grayscale (source not vendored)."""
import ctypes

import numpy as np
import torch

from . import d2


def _lib():
    lib = d2._lib()
    if not getattr(lib, "_aug_ready", False):
        vp, i64, f, i = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
        for name, args in (("ptaug_to_gray", [vp, vp, i64]), ("ptaug_brightness", [vp, vp, i64, f]),
                           ("ptaug_contrast", [vp, vp, i64, f]), ("ptaug_saturation", [vp, vp, i64, f]),
                           ("ptaug_hue", [vp, vp, i64, i]), ("ptaug_solarize", [vp, vp, i64, i]),
                           ("ptaug_gaussian_blur", [vp, vp, i, i, f])):
            getattr(lib, name).restype = None
            getattr(lib, name).argtypes = args
        lib.ptaug_box_weights.restype = None
        lib.ptaug_box_weights.argtypes = [f, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint32),
                                          ctypes.POINTER(ctypes.c_uint32)]
        lib.ptaug_resize_bilinear.restype = None
        lib.ptaug_resize_bilinear.argtypes = [vp, vp, i, i, i, i, i]
        lib.ptaug_gray_mean.restype = i
        lib.ptaug_gray_mean.argtypes = [vp, i64]
        lib._aug_ready = True
    return lib


def _run(name, img: torch.Tensor, *args):
    img = img.contiguous()
    assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[0] == 3
    out = torch.empty_like(img)
    hw = img.shape[1] * img.shape[2]
    if name == "ptaug_gaussian_blur":
        _lib().ptaug_gaussian_blur(img.data_ptr(), out.data_ptr(), img.shape[1], img.shape[2], *args)
    else:
        getattr(_lib(), name)(img.data_ptr(), out.data_ptr(), hw, *args)
    return out


def c_gray(img): return _run("ptaug_to_gray", img)
def c_brightness(img, f): return _run("ptaug_brightness", img, float(f))
def c_contrast(img, f): return _run("ptaug_contrast", img, float(f))
def c_saturation(img, f): return _run("ptaug_saturation", img, float(f))
def hue_shift(hue_factor: float) -> int:
    """`np.uint8(hue_factor * 255)` as torchvision 0.8.2 evaluates it under the numpy of its day: the C float -> uint8
    conversion, i.e. truncation toward zero, then modulo 256 (numpy >= 2 raises for negative values instead)."""
    return int(hue_factor * 255) & 0xFF


def c_hue(img, hue_factor): return _run("ptaug_hue", img, hue_shift(hue_factor))
def c_solarize(img, thr=128): return _run("ptaug_solarize", img, int(thr))
def c_blur(img, sigma): return _run("ptaug_gaussian_blur", img, float(sigma))


def c_resize(img: torch.Tensor, nh: int, nw: int) -> torch.Tensor:
    """Image.resize((nw, nh), Image.BILINEAR) on a planar uint8 (C, H, W) image (D2 ResizeTransform.apply_image)"""
    img = img.contiguous()
    assert img.dtype == torch.uint8 and img.dim() == 3
    out = torch.empty((img.shape[0], nh, nw), dtype=torch.uint8)
    _lib().ptaug_resize_bilinear(img.data_ptr(), out.data_ptr(), img.shape[0], img.shape[1], img.shape[2], nh, nw)
    return out


def pil_resize(img: torch.Tensor, nh: int, nw: int) -> torch.Tensor:
    from PIL import Image
    return _from_pil(_to_pil(img).resize((nw, nh), Image.BILINEAR))


def c_box_weights(sigma: float):
    r, ww, fw = ctypes.c_int(), ctypes.c_uint32(), ctypes.c_uint32()
    _lib().ptaug_box_weights(float(sigma), ctypes.byref(r), ctypes.byref(ww), ctypes.byref(fw))
    return r.value, ww.value, fw.value


def apply_strong(img: torch.Tensor, p) -> torch.Tensor:
    """the whole strong pipeline for one image and one parameter draw `p` (fields jitter / gray / blur_sigma / solarize
    as probabilisticteacher_amd.data.StrongParams; ops 1 brightness, 2 contrast, 3 saturation, 4 hue), C restatement"""
    fns = {1: c_brightness, 2: c_contrast, 3: c_saturation, 4: c_hue}
    for op, f in p.jitter:
        img = fns[op](img, f)
    if p.gray:
        img = c_gray(img)
    if p.blur_sigma is not None:
        img = c_blur(img, p.blur_sigma)
    if p.solarize is not None:
        img = c_solarize(img, p.solarize)
    return img


# ---------------------------------------------------------------- the literal PIL calls
def _to_pil(img: torch.Tensor):
    from PIL import Image
    return Image.fromarray(np.ascontiguousarray(img.numpy().transpose(1, 2, 0)), "RGB")     # dataset_mapper.py:155


def _from_pil(pil) -> torch.Tensor:
    return torch.as_tensor(np.ascontiguousarray(np.array(pil).transpose(2, 0, 1)))           # dataset_mapper.py:156-159


def pil_brightness(img, f):
    from PIL import ImageEnhance
    return _from_pil(ImageEnhance.Brightness(_to_pil(img)).enhance(f))


def pil_contrast(img, f):
    from PIL import ImageEnhance
    return _from_pil(ImageEnhance.Contrast(_to_pil(img)).enhance(f))


def pil_saturation(img, f):
    from PIL import ImageEnhance
    return _from_pil(ImageEnhance.Color(_to_pil(img)).enhance(f))


def pil_hue(img, hue_factor):
    from PIL import Image
    h, s, v = _to_pil(img).convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h += np.uint8(hue_shift(hue_factor))
    return _from_pil(Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB"))


def pil_gray(img):
    from PIL import Image
    l = np.array(_to_pil(img).convert("L"), dtype=np.uint8)
    return _from_pil(Image.fromarray(np.dstack([l, l, l]), "RGB"))


def pil_solarize(img, thr=128):
    from PIL import ImageOps
    return _from_pil(ImageOps.solarize(_to_pil(img), thr))


def pil_blur(img, sigma):
    from PIL import ImageFilter
    return _from_pil(_to_pil(img).filter(ImageFilter.GaussianBlur(radius=sigma)))
