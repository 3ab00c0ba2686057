"""Alternate slicing modes: the live part of the reference's ``llava/process_image.py``
(``any_res`` and ``pad_then_devide`` branches of ``process_images``; also imported by train.py:39).

Restated from the algorithm (process_image.py:70-101,119-139,156-167,189-214).  Helpers of the
reference's commented-out code (torch_extract_patches, adapt_size, sliding_window, ...) are dead and
not reproduced.  Host code, PIL only (no torchvision dependency).
"""
from __future__ import annotations

import math
from typing import List, Tuple

from PIL import Image

from .constants import IMAGE_HEIGHT, IMAGE_WIDTH, PATCH_SIZE, PATCH_NUM_WIDTH, PATCH_NUM_HEIGHT  # noqa: F401

MAX_PATCHES = PATCH_NUM_WIDTH * PATCH_NUM_HEIGHT
TOKEN_LENGTH = 3 * PATCH_SIZE * PATCH_SIZE
POSITION_EMBEDDING_LENGTH = 1024


def _factor_triples(n: int) -> List[Tuple[float, int, int]]:
    return [(i / (n / i), i, n // i) for i in range(1, n + 1) if n % i == 0]


def cal_num_of_slices(origin_image_width: int, origin_image_height: int) -> Tuple[int, int]:
    """(w_slices, h_slices): slice count = ceil(area/336^2) capped at 6 (no 1->2 bump here, unlike the
    anyres rule); among factorizations of the neighbouring counts pick the aspect ratio closest in log
    space, first wins (process_image.py:70-101)."""
    scale = min(math.ceil(origin_image_width * origin_image_height / (IMAGE_WIDTH * IMAGE_HEIGHT)), 6)
    if scale <= 2:
        cands = _factor_triples(scale) + _factor_triples(scale + 1)
    else:
        cands = _factor_triples(scale - 1) + _factor_triples(scale) + _factor_triples(scale + 1)
    target = math.log(origin_image_width / origin_image_height)
    best_w = best_h = 0
    best = 1000
    for r, w, h in cands:
        d = abs(math.log(r) - target)
        if best > d:
            best, best_w, best_h = d, w, h
    return best_w, best_h


def slice_image_any_res(image: Image.Image) -> List[Image.Image]:
    """Integer-box slices, row-major (process_image.py:119-139)."""
    W, H = image.size
    bw, bh = cal_num_of_slices(W, H)
    return [image.crop((i * W // bw, j * H // bh, (i + 1) * W // bw, (j + 1) * H // bh)).convert("RGB")
            for j in range(bh) for i in range(bw)]


def expand2square(pil_img: Image.Image, background_color) -> Image.Image:
    """Centre the image on a square canvas of the longer side (process_image.py:156-167)."""
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    out = Image.new(pil_img.mode, (side, side), background_color)
    out.paste(pil_img, (0, (w - h) // 2) if w > h else ((h - w) // 2, 0))
    return out


def resize_image(image: Image.Image, target_width: int) -> Image.Image:
    """LANCZOS resize to a target width, height truncated (process_image.py:189-193)."""
    ratio = target_width / float(image.size[0])
    return image.resize((target_width, int(float(image.size[1]) * float(ratio))), Image.LANCZOS)


def _windows(image: Image.Image, window: Tuple[int, int], stride: int) -> List[Image.Image]:
    w, h = image.size
    return [image.crop((x, y, x + window[0], y + window[1]))
            for y in range(0, h - window[1] + 1, stride) for x in range(0, w - window[0] + 1, stride)]


def process_image_any_res(image: Image.Image, background_color=0) -> List[Image.Image]:
    """[whole image] + slices, each squared with the background colour; sizes vary, the image
    processor resizes them later (process_image.py:195-202)."""
    image = image.convert("RGB")
    return [expand2square(v, background_color) for v in [image] + slice_image_any_res(image)]


def process_image_naive(image: Image.Image, background_color=0) -> List[Image.Image]:
    """Squared image + 336^2 windows at stride 308 of its 1024-wide LANCZOS resize: 1 + 9 views
    (process_image.py:204-214)."""
    image = expand2square(image, background_color)
    return [image] + _windows(resize_image(image, 1024), (IMAGE_WIDTH, IMAGE_HEIGHT), 308)
