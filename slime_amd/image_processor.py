"""Minimal CLIP image processor with the attribute surface the reference's ``mm_utils`` touches
(``crop_size['height']``, ``crop_size.values()``, ``size['shortest_edge']``, ``image_mean``,
``preprocess(img, return_tensors='pt')['pixel_values']``) -- the role HF ``CLIPImageProcessor`` plays at
llava/model/multimodal_encoder/clip_encoder.py:30.  Host (CPU, PIL + numpy) code: it runs inside
forked DataLoader workers in the reference (llava/eval/model_vqa_loader.py:78), where HIP cannot.

Arithmetic follows transformers' PIL backend so outputs are bit-identical to it:
resize shortest edge (bicubic, long side ``int(size*long/short)``) -> centre crop (``(h-crop)//2``) ->
``float32(float64(u8) * (1/255))`` -> ``(x - mean32) / std32`` -> CHW.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Sequence, Union

import numpy as np
import torch
from PIL import Image

from .constants import OPENAI_CLIP_MEAN, OPENAI_CLIP_STD


class ClipImageProcessor:
    def __init__(self, size: int = 336, crop_size: int = 336, image_mean: Sequence[float] = OPENAI_CLIP_MEAN,
                 image_std: Sequence[float] = OPENAI_CLIP_STD, rescale_factor: float = 1 / 255):
        self.size: Dict[str, int] = {"shortest_edge": int(size)}
        self.crop_size: Dict[str, int] = {"height": int(crop_size), "width": int(crop_size)}
        self.image_mean: List[float] = [float(v) for v in image_mean]
        self.image_std: List[float] = [float(v) for v in image_std]
        self.rescale_factor = float(rescale_factor)
        self.resample = Image.BICUBIC

    @classmethod
    def from_pretrained(cls, path: str) -> "ClipImageProcessor":
        """Read an HF ``preprocessor_config.json`` from a local directory (no hub access here)."""
        f = os.path.join(path, "preprocessor_config.json")
        if not os.path.isfile(f):
            return cls()
        cfg = json.load(open(f))
        size = cfg.get("size", 336)
        size = size.get("shortest_edge", size.get("height", 336)) if isinstance(size, dict) else size
        crop = cfg.get("crop_size", size)
        crop = crop.get("height", size) if isinstance(crop, dict) else crop
        return cls(size, crop, cfg.get("image_mean", OPENAI_CLIP_MEAN), cfg.get("image_std", OPENAI_CLIP_STD),
                   cfg.get("rescale_factor", 1 / 255))

    # -- pieces -------------------------------------------------------------------------------
    def _resize_center_crop(self, img: Image.Image) -> np.ndarray:
        img = img.convert("RGB")
        w, h = img.size
        s = self.size["shortest_edge"]
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = s, int(s * long / short)
        nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        if (nw, nh) != (w, h):
            img = img.resize((nw, nh), resample=self.resample)
        arr = np.asarray(img)                                  # HWC uint8
        ch, cw = self.crop_size["height"], self.crop_size["width"]
        top, left = (nh - ch) // 2, (nw - cw) // 2
        if top < 0 or left < 0:                                # smaller than the crop: zero pad (HF center_crop)
            ph, pw = max(ch, nh), max(cw, nw)
            pad = np.zeros((ph, pw, 3), dtype=arr.dtype)
            t0, l0 = -(-(ph - nh) // 2), -(-(pw - nw) // 2)
            pad[t0:t0 + nh, l0:l0 + nw] = arr
            arr, top, left = pad, (ph - ch) // 2, (pw - cw) // 2
        return arr[top:top + ch, left:left + cw]

    def normalize_u8(self, arr: np.ndarray) -> np.ndarray:
        """HWC uint8 -> CHW float32, HF numerics."""
        x = (arr.astype(np.float64) * self.rescale_factor).astype(np.float32)
        x = (x - np.array(self.image_mean, dtype=np.float32)) / np.array(self.image_std, dtype=np.float32)
        return np.ascontiguousarray(x.transpose(2, 0, 1))

    def preprocess(self, images: Union[Image.Image, Sequence[Image.Image]], return_tensors: str = "pt"):
        if isinstance(images, Image.Image):
            images = [images]
        out = [self.normalize_u8(self._resize_center_crop(im)) for im in images]
        if return_tensors == "pt":
            return {"pixel_values": torch.from_numpy(np.stack(out, 0))}
        return {"pixel_values": out}

    __call__ = preprocess
