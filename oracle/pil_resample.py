"""TEST INFRASTRUCTURE (oracle): numpy restatement of Pillow's 8-bit separable resize.

The reference's slicer calls ``PIL.Image.resize`` with the default filter (llava/mm_utils.py:117 in
``resize_and_pad_image`` and :200 for the global thumbnail); Pillow is a third-party dependency that is
NOT under /root/reference (pinned only as ``Pillow`` by the reference; 12.2.0 is what this image and the
GPU box carry).  The algorithm restated here is Pillow's published ``ImagingResample`` for 8 bpc images
(src/libImaging/Resample.c: ``precompute_coeffs``, ``normalize_coeffs_8bpc``, ``ImagingResampleHorizontal_8bpc``,
``ImagingResampleVertical_8bpc``): a two-pass separable convolution -- horizontal first, rounded to uint8,
then vertical -- whose double-precision filter weights are normalised per output pixel and converted to
22-bit fixed point.  Pinned by tests/test_oracle_golden.py against Pillow itself (the installed library is
the golden source) over up/down/mixed scales.  Only tests/ may import this file.
"""
from __future__ import annotations

import math
import numpy as np

PRECISION_BITS = 32 - 8 - 2
BICUBIC_SUPPORT = 2.0


def _bicubic(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def coefficients(in_size: int, out_size: int):
    """(xmin [out], count [out], fixed-point weights [out, ksize]) for the full box (0, in_size)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = BICUBIC_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    centers = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((centers - support + 0.5).astype(np.int64), 0)         # C cast: truncation
    xmax = np.minimum((centers + support + 0.5).astype(np.int64), in_size)
    count = xmax - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    w = _bicubic((taps + xmin[:, None] - centers[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(taps < count[:, None], w, 0.0)
    # Pillow accumulates ww sequentially over the taps; numpy's row sum may associate differently, so
    # do the accumulation in the same order
    ww = np.zeros(out_size, dtype=np.float64)
    for t in range(ksize):
        ww = ww + w[:, t]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    fixed = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)), (0.5 + w * (1 << PRECISION_BITS))).astype(np.int64)
    return xmin, count, fixed   # astype truncates toward zero like the C cast


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """Resample ``img`` [H, W, C] uint8 along ``axis`` (0 = vertical, 1 = horizontal)."""
    in_size = img.shape[axis]
    xmin, count, kk = coefficients(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)                          # [in, other, C]
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for o in range(out_size):
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t in range(int(count[o])):
            acc += src[xmin[o] + t] * kk[o, t]
        out[o] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``np.asarray(PIL.Image.fromarray(img).resize((out_w, out_h)))`` for an RGB uint8 array."""
    h, w, _ = img.shape
    if w != out_w:
        img = _pass(img, out_w, 1)
    if h != out_h:
        img = _pass(img, out_h, 0)
    return img.copy()


def resize_and_pad_u8(img: np.ndarray, target_w: int, target_h: int) -> np.ndarray:
    """uint8 restatement of ``resize_and_pad_image`` (mm_utils.py:99-131): aspect-preserving resize (ceil on
    the free side) pasted centred on a black canvas."""
    oh, ow, _ = img.shape
    sw, sh = target_w / ow, target_h / oh
    if sw < sh:
        nw, nh = target_w, min(math.ceil(oh * sw), target_h)
    else:
        nh, nw = target_h, min(math.ceil(ow * sh), target_w)
    canvas = np.zeros((target_h, target_w, 3), dtype=np.uint8)
    x0, y0 = (target_w - nw) // 2, (target_h - nh) // 2
    canvas[y0:y0 + nh, x0:x0 + nw] = resize_bicubic_u8(img, nw, nh)
    return canvas
