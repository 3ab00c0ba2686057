"""Checkpoint ingestion for the visual hot path: the adapter-loading half of llava/model/builder.py:64-109 and
llava/model/llava_arch.py:107-119 (the LLM / tokenizer / LoRA half of ``load_pretrained_model`` is outside SURVEY.md section 8).

The reference stores the trained adapter in up to three torch-pickle files next to a checkpoint:

  ``mm_projector.bin``          pretrain stage (train.py ``safe_save_model_for_hf_trainer``): keys ``model.mm_projector.*``,
                                loaded with ``model.load_state_dict(..., strict=False)`` after a cast to fp16 (builder.py:106-108)
                                or through ``get_w(weights, 'mm_projector')`` = text after ``'mm_projector.'`` (llava_arch.py:107-112)
  ``sampler.bin``               same for ``model.sampler.*`` (llava_arch.py:114-119)
  ``non_lora_trainables.bin``   LoRA finetune: keys ``base_model.model.model.mm_projector.*`` / ``...sampler.*``; the reference strips
                                ``base_model.`` and then one ``model.`` (builder.py:93-96)

``read_adapter_state`` accepts any of the files (or a directory holding them, or ``.safetensors`` equivalents) and returns one
dict keyed ``mm_projector.*`` / ``sampler.*`` whatever the prefix was; ``load_adapter_checkpoint`` puts it into a model built by
``build_vision_projector`` / ``build_vision_sampler``.  Parameter dtypes follow the reference: ``w_gate`` / ``w_noise`` stay bf16
Parameters, ``pos_embed`` fp16, everything else takes the checkpoint's values in the module's dtype.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional

import torch

ADAPTER_FILES = ("mm_projector.bin", "sampler.bin", "non_lora_trainables.bin",
                 "mm_projector.safetensors", "sampler.safetensors", "non_lora_trainables.safetensors")
_KEYWORDS = ("mm_projector", "sampler")


def _read_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(sd, dict):
        raise ValueError(f"{path}: expected a state dict")
    return sd


def canonical_adapter_key(key: str) -> Optional[str]:
    """``[base_model.][model.]*mm_projector.X`` -> ``mm_projector.X`` (same for ``sampler``); None for foreign keys (LoRA /
    LLM tensors of non_lora_trainables.bin such as embed_tokens).  The reference's ``get_w`` keeps the text after the first
    ``'<keyword>.'`` (llava_arch.py:108-109); anchoring at a dot boundary keeps ``...resampler.`` from matching ``sampler.``."""
    parts = key.split(".")
    for i, p in enumerate(parts[:-1]):
        if p in _KEYWORDS:
            return ".".join(parts[i:])
    return None


def read_adapter_state(path: str, files: Iterable[str] = ADAPTER_FILES) -> Dict[str, torch.Tensor]:
    paths = [path] if os.path.isfile(path) else [os.path.join(path, f) for f in files if os.path.isfile(os.path.join(path, f))]
    if not paths:
        raise FileNotFoundError(f"no adapter checkpoint at {path!r} (looked for {', '.join(files)})")
    out: Dict[str, torch.Tensor] = {}
    for p in paths:
        for k, v in _read_file(p).items():
            ck = canonical_adapter_key(k)
            if ck is not None:
                out[ck] = v
    if not out:
        raise ValueError(f"{paths}: no mm_projector.* / sampler.* tensors found")
    return out


def load_adapter_checkpoint(model, path: str, strict: bool = True) -> Dict[str, torch.Tensor]:
    """Load ``mm_projector`` / ``sampler`` of ``model`` (anything exposing them, or ``get_model()``) from the files at ``path``.
    ``strict``: every parameter of a module that appears in the checkpoint must be present and nothing may be left over
    (the reference loads with strict=False and would run missing tensors at their random initialisation)."""
    m = model.get_model() if hasattr(model, "get_model") else model
    state = read_adapter_state(path)
    for kw in _KEYWORDS:
        mod = getattr(m, kw, None)
        sub = {k[len(kw) + 1:]: v for k, v in state.items() if k.startswith(kw + ".")}
        if not sub:
            continue
        if mod is None or len(list(mod.state_dict().keys())) == 0:
            if strict:
                raise RuntimeError(f"checkpoint holds {len(sub)} {kw}.* tensors but the model has no such module")
            continue
        own = mod.state_dict()
        cast = {k: v.to(own[k].dtype) if k in own and v.is_floating_point() else v for k, v in sub.items()}
        res = mod.load_state_dict(cast, strict=False)
        if strict and (res.missing_keys or res.unexpected_keys):
            raise RuntimeError(f"{kw}: missing {res.missing_keys[:4]} unexpected {res.unexpected_keys[:4]}")
    return state
