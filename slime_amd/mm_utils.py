"""Image slicer and pre-processing front end: the ``llava.mm_utils`` surface of the hot path.

Same function names, arguments and behaviour as the reference (llava/mm_utils.py:14-259) so drivers
can switch imports; written from the algorithm, not from the text.  This is host code (PIL): in
the reference it runs inside forked DataLoader workers, where HIP is unavailable.  The on-device
counterpart (``process_anyres_image_gpu``) runs resize + pad + tile + normalise in HIP, bit-identically.
"""
from __future__ import annotations

import ast
import math
from typing import List, Sequence, Tuple, Union

import numpy as np
import torch
from PIL import Image

from .process_image import process_image_naive, process_image_any_res, expand2square  # noqa: F401 (API surface)

CROP = 336   # the reference hard-codes (336, 336) in get_anyres_image_grid_shape (mm_utils.py:173)


def _parse_pinpoints(grid_pinpoints) -> List[Tuple[int, int]]:
    return grid_pinpoints if type(grid_pinpoints) is list else ast.literal_eval(grid_pinpoints)


def select_best_resolution(original_size: Tuple[int, int], possible_resolutions: Sequence[Tuple[int, int]]):
    """LLaVA-1.6 pinpoint choice (mm_utils.py:12-39): max effective resolution, tie -> min waste.
    Kept for API completeness; SliME overrides its result (see get_anyres_image_grid_shape)."""
    ow, oh = original_size
    best, best_eff, best_waste = None, 0, float("inf")
    for w, h in possible_resolutions:
        s = min(w / ow, h / oh)
        eff = min(int(ow * s) * int(oh * s), ow * oh)
        waste = w * h - eff
        if eff > best_eff or (eff == best_eff and waste < best_waste):
            best, best_eff, best_waste = (w, h), eff, waste
    return best


def _slice_candidates(scale: int) -> List[Tuple[int, int]]:
    """(w_slices, h_slices) factor pairs of the slice counts adjacent to ``scale`` (mm_utils.py:66-81)."""
    def pairs(n):
        return [(i, n // i) for i in range(1, n + 1) if n % i == 0]
    if scale <= 2:
        return pairs(scale) + pairs(scale + 1)
    return pairs(scale - 1) + pairs(scale) + pairs(scale + 1)


def select_best_resolution_uhd(original_size: Tuple[int, int], processor_size) -> Tuple[int, int]:
    """SliME's adaptive slicing rule (mm_utils.py:41-97): slice count = ceil(area / crop area), capped
    at 6, a single slice is bumped to 2; among the factorizations of the neighbouring slice counts keep
    the canvas with the largest effective resolution (ties: least padding, first wins)."""
    iw, ih = tuple(processor_size)
    ow, oh = original_size
    scale = min(math.ceil(ow * oh / (iw * ih)), 6)
    if scale == 1:
        scale = 2
    best, best_eff, best_waste = None, 0, float("inf")
    for ws, hs in _slice_candidates(scale):
        w, h = ws * iw, hs * ih
        s = min(w / ow, h / oh)
        eff = min(int(ow * s) * int(oh * s), ow * oh)
        waste = w * h - eff
        if eff > best_eff or (eff == best_eff and waste < best_waste):
            best, best_eff, best_waste = (w, h), eff, waste
    return best


def resize_and_pad_image(image: Image.Image, target_resolution: Tuple[int, int]) -> Image.Image:
    """Aspect-preserving resize (ceil on the free side, PIL default resample = bicubic) centred on a
    black canvas (mm_utils.py:99-131)."""
    ow, oh = image.size
    tw, th = target_resolution
    sw, sh = tw / ow, th / oh
    if sw < sh:
        nw, nh = tw, min(math.ceil(oh * sw), th)
    else:
        nh, nw = th, min(math.ceil(ow * sh), tw)
    canvas = Image.new("RGB", (tw, th), (0, 0, 0))
    canvas.paste(image.resize((nw, nh)), ((tw - nw) // 2, (th - nh) // 2))
    return canvas


def divide_to_patches(image: Image.Image, patch_size: int) -> List[Image.Image]:
    """Row-major tiling, height outer (mm_utils.py:134-153)."""
    w, h = image.size
    return [image.crop((x, y, x + patch_size, y + patch_size))
            for y in range(0, h, patch_size) for x in range(0, w, patch_size)]


def get_anyres_image_grid_shape(image_size, grid_pinpoints, patch_size) -> Tuple[int, int]:
    """(num_patch_width, num_patch_height) of the local-crop grid (mm_utils.py:156-174).  The reference
    evaluates the pinpoint rule and then overwrites it with the uhd rule for a hard-coded 336 crop,
    so ``grid_pinpoints`` has no effect on the result; it is still parsed (and may raise) as there."""
    select_best_resolution(image_size, _parse_pinpoints(grid_pinpoints))
    w, h = select_best_resolution_uhd(image_size, (CROP, CROP))
    return w // patch_size, h // patch_size


def anyres_canvas(image: Image.Image, processor) -> Tuple[Image.Image, Image.Image]:
    """(global 336x336 thumbnail -- aspect NOT preserved, mm_utils.py:200 --, padded local canvas)."""
    best = select_best_resolution_uhd(image.size, tuple(processor.crop_size.values()))
    canvas = resize_and_pad_image(image, best)
    s = processor.size["shortest_edge"]
    return image.resize((s, s)), canvas


def process_anyres_image(image: Image.Image, processor, grid_pinpoints) -> torch.Tensor:
    """[1+n, 3, 336, 336] fp32: global view first, then the row-major tiles (mm_utils.py:177-210)."""
    _parse_pinpoints(grid_pinpoints)
    thumb, canvas = anyres_canvas(image, processor)
    views = [thumb] + divide_to_patches(canvas, processor.crop_size["height"])
    return torch.stack([processor.preprocess(v, return_tensors="pt")["pixel_values"][0] for v in views], dim=0)


def slice_image_gpu(image_u8: torch.Tensor, crop: int = CROP) -> Tuple[torch.Tensor, torch.Tensor]:
    """The slicer on the device: uint8 [H, W, 3] image (already in HBM) -> (uint8 [crop, crop, 3] global
    thumbnail, uint8 padded local canvas), bit-identical to :func:`anyres_canvas` -- the uhd grid choice on
    the host (integers), both bicubic resamplings in ``slime_resize_bicubic_u8`` (Pillow's fixed-point
    arithmetic), the centred paste as the destination view of the second one."""
    from . import ops
    H, W, _ = image_u8.shape
    tw, th = select_best_resolution_uhd((W, H), (crop, crop))
    sw, sh = tw / W, th / H
    if sw < sh:
        nw, nh = tw, min(math.ceil(H * sw), th)
    else:
        nh, nw = th, min(math.ceil(W * sh), tw)
    thumb = ops.resize_bicubic_u8(image_u8, crop, crop)
    canvas = torch.zeros((th, tw, 3), dtype=torch.uint8, device=image_u8.device)
    x0, y0 = (tw - nw) // 2, (th - nh) // 2
    ops.resize_bicubic_u8(image_u8, nw, nh, out=canvas[y0:y0 + nh, x0:x0 + nw])
    return thumb, canvas


def process_anyres_image_gpu(image, processor, grid_pinpoints, device, dtype=torch.float32) -> torch.Tensor:
    """Same result as :func:`process_anyres_image` (bit-identical in fp32) with the whole slicer on the GPU:
    only the raw uint8 pixels cross PCIe; resize + pad (``slime_resize_bicubic_u8``), tiling, rescale and
    normalise (``slime_tile_normalize``) run in HIP.  ``image``: PIL image or uint8 [H, W, 3] tensor.
    Must be called from the main process (HIP is unusable in forked DataLoader workers)."""
    from . import ops
    _parse_pinpoints(grid_pinpoints)
    if isinstance(image, Image.Image):
        image = torch.from_numpy(np.array(image.convert("RGB")))
    image = image.to(device, non_blocking=True)
    crop = processor.crop_size["height"]
    thumb, canvas = slice_image_gpu(image, crop)
    g = ops.tile_normalize(thumb, crop, processor.image_mean, processor.image_std, dtype)
    l = ops.tile_normalize(canvas, crop, processor.image_mean, processor.image_std, dtype)
    return torch.cat([g, l], dim=0)


def process_images_gpu(images, image_processor, model_cfg, device, dtype=torch.float32):
    """Device counterpart of :func:`process_images` for the SliME mode (``image_aspect_ratio='anyres'``):
    images (PIL or uint8 [H, W, 3] tensors, host or device) -> the same stacked tensor / list, resident on
    ``device``.  Other modes raise: they are API-completeness modes of the reference served by the host path."""
    mode = getattr(model_cfg, "image_aspect_ratio", None)
    if mode != "anyres":
        raise NotImplementedError(f"process_images_gpu implements image_aspect_ratio='anyres' only (got {mode!r}); "
                                  "use process_images for the other modes")
    _parse_pinpoints(model_cfg.image_grid_pinpoints)
    arrs = [torch.from_numpy(np.array(im.convert("RGB"))) if isinstance(im, Image.Image) else im for im in images]
    if len(arrs) > 1 and all(a.shape == arrs[0].shape for a in arrs):
        # one resolution bucket: the whole batch goes through the slicer in 6 launches (2 resizes x 2 passes, 2 tile+normalise)
        from . import ops
        batch = torch.empty((len(arrs),) + tuple(arrs[0].shape), dtype=torch.uint8, device=device)   # [B, H, W, 3]
        for i, a in enumerate(arrs):
            batch[i].copy_(a, non_blocking=True)            # straight from the caller's (ideally pinned) buffers
        B, H, W, _ = batch.shape
        crop = image_processor.crop_size["height"]
        tw, th = select_best_resolution_uhd((W, H), (crop, crop))
        sw, sh = tw / W, th / H
        if sw < sh:
            nw, nh = tw, min(math.ceil(H * sw), th)
        else:
            nh, nw = th, min(math.ceil(W * sh), tw)
        n_local = (tw // crop) * (th // crop)
        thumbs = ops.resize_bicubic_u8_batched(batch, crop, crop)
        if (nw, nh) == (W, H) == (tw, th):
            canvas = batch                                 # already a whole number of crops: Pillow's resize is a copy
        else:
            canvas = torch.zeros((B, th, tw, 3), dtype=torch.uint8, device=batch.device)
            x0, y0 = (tw - nw) // 2, (th - nh) // 2
            ops.resize_bicubic_u8_batched(batch, nw, nh, out=canvas[:, y0:y0 + nh, x0:x0 + nw])
        out = torch.empty((B, 1 + n_local, 3, crop, crop), dtype=dtype, device=batch.device)
        ops.tile_normalize_batched(thumbs, crop, image_processor.image_mean, image_processor.image_std, out, 0)
        ops.tile_normalize_batched(canvas, crop, image_processor.image_mean, image_processor.image_std, out, 1)
        return out
    out = [process_anyres_image_gpu(a, image_processor, model_cfg.image_grid_pinpoints, device, dtype) for a in arrs]
    if all(x.shape == out[0].shape for x in out):
        return torch.stack(out, dim=0)
    return out


def process_images(images: Sequence[Image.Image], image_processor, model_cfg) -> Union[torch.Tensor, List[torch.Tensor]]:
    """Mode switch on ``model_cfg.image_aspect_ratio`` (mm_utils.py:231-259): 'pad' (1 crop),
    'pad_then_devide' (1+9), 'any_res' (1+slices, variable sizes resized by the processor), 'anyres'
    (SliME), else the bare processor.  Stacks to one tensor when all per-image shapes agree."""
    mode = getattr(model_cfg, "image_aspect_ratio", None)
    bg = tuple(int(x * 255) for x in image_processor.image_mean)

    def pp(img):
        return image_processor.preprocess(img, return_tensors="pt")["pixel_values"][0]

    out = []
    if mode == "pad":
        out = [pp(expand2square(im, bg)) for im in images]
    elif mode == "pad_then_devide":
        out = [torch.stack([pp(v) for v in process_image_naive(im, bg)]) for im in images]
    elif mode == "any_res":
        out = [torch.stack([pp(v) for v in process_image_any_res(im, bg)]) for im in images]
    elif mode == "anyres":
        out = [process_anyres_image(im, image_processor, model_cfg.image_grid_pinpoints) for im in images]
    else:
        return image_processor(list(images), return_tensors="pt")["pixel_values"]
    if all(x.shape == out[0].shape for x in out):
        return torch.stack(out, dim=0)
    return out
