"""``build_vision_projector`` and the gated global adapter (llava/model/multimodal_projector/builder.py).

``GatedBlock`` keeps the reference's parameter names (``projection.{0,2}.*``, ``attn.*``, ``w_gate``,
``w_noise``, buffers ``mean``/``std``) and its forward contract (:179-209): inputs whose dim0 and dim1 are
both != 576 return ``projection(x)``; otherwise the softmax-gated mix of ``projection(x)`` and
``projection(attn(x))`` (or one expert when ``mm_learnable_gated >= 0``).  All arithmetic runs in
libslime_hip (GEMM with fused erf-GELU epilogue, Resampler kernels, gate-mix kernel).  Inference
only: the training-time noisy gating / load-balancing terms (:76-176) are outside the hot path.
"""
from __future__ import annotations

import math
import re
from typing import Dict, Optional

import torch
import torch.nn as nn

from ..multimodal_resampler.sampler import Resampler
from ... import ops
from ...weights import AdapterConfig


class IdentityMap(nn.Module):
    def forward(self, x, *args, **kwargs):
        return x

    @property
    def config(self):
        return {"mm_projector_type": "identity"}


def _operand_dtype(x: torch.Tensor, w: torch.Tensor, default=torch.bfloat16) -> torch.dtype:
    for t in (x, w):
        if t.dtype in (torch.bfloat16, torch.float16):
            return t.dtype
    return default


class HipMlp(nn.Sequential):
    """``mlpNx_gelu`` (N = 2: the 'projection' of SliME) = Linear -> exact GELU -> Linear, state-dict keys
    ``0.*`` / ``2.*`` as the reference's nn.Sequential (projector/builder.py:241-248)."""

    def __init__(self, in_dim: int, hidden: int, depth: int = 2):
        if depth != 2:
            raise NotImplementedError("only mlp2x_gelu is on the SliME hot path")
        super().__init__(nn.Linear(in_dim, hidden), nn.GELU(), nn.Linear(hidden, hidden))
        self._packed: Dict = {}

    def _apply(self, fn, *a, **kw):
        self._packed.clear()
        return super()._apply(fn, *a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        self._packed.clear()
        return super()._load_from_state_dict(*a, **kw)

    def packed(self, dtype: torch.dtype) -> ops.PackedMlp:
        key = (dtype, str(self[0].weight.device))
        if key not in self._packed:
            self._packed[key] = ops.pack_mlp(self[0].weight, self[0].bias, self[2].weight, self[2].bias, dtype,
                                             self[0].weight.device)
        return self._packed[key]

    @torch.no_grad()
    def forward(self, x, out_dtype: Optional[torch.dtype] = None, operand_dtype: Optional[torch.dtype] = None):
        """``operand_dtype`` (extension): the 16-bit MFMA operand type when ``x`` is handed over in fp32 -- ``encode_images`` passes
        the IMAGES' 16-bit type (the type the reference's adapter computes in: its tower returns features in the input dtype,
        clip_encoder.py:52,56); default: x's own 16-bit type, else the weights', else bf16."""
        shp = x.shape
        pm = self.packed(operand_dtype or _operand_dtype(x, self[0].weight))
        out = ops.mlp_forward(pm, x.reshape(-1, shp[-1]))
        return out.view(*shp[:-1], -1).to(out_dtype or x.dtype)


class HipLinear(nn.Linear):
    """``mm_projector_type='linear'``: one GEMM with bias."""

    @torch.no_grad()
    def forward(self, x, out_dtype: Optional[torch.dtype] = None, operand_dtype: Optional[torch.dtype] = None):
        """``out_dtype`` / ``operand_dtype``: the extensions encode_images passes to every projector type (see HipMlp.forward)."""
        from ... import _lib
        dt = operand_dtype or _operand_dtype(x, self.weight)
        shp = x.shape
        a = x.reshape(-1, shp[-1]).to(dt).contiguous()
        out = ops.gemm(a, self.weight.to(dt).contiguous(), self.bias.float().contiguous(), _lib.EPI_BIAS_F32)
        return out.view(*shp[:-1], -1).to(out_dtype or x.dtype)


class GatedBlock(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.target_sequence_length = 576
        grid = int(math.sqrt(self.target_sequence_length))
        self.attn = Resampler(grid_size=grid, embed_dim=config.mm_hidden_size, num_heads=config.mm_hidden_size // 128,
                              kv_dim=config.mm_hidden_size, llm_hidden_size=config.hidden_size, use_post_proj=False)
        self.projection = HipMlp(config.mm_hidden_size, config.hidden_size)
        self.expert_ffn = [self.projection, self.attn]          # attribute train.py:1124 touches
        self.num_experts = 2
        self.w_gate = nn.Parameter(torch.zeros(config.mm_hidden_size, 2, dtype=torch.bfloat16))
        self.w_noise = nn.Parameter(torch.zeros(config.mm_hidden_size, 2, dtype=torch.bfloat16))
        self.register_buffer("mean", torch.tensor([0.0], dtype=torch.bfloat16))
        self.register_buffer("std", torch.tensor([1.0], dtype=torch.bfloat16))
        self.learnable_gated = getattr(config, "mm_learnable_gated", -1)
        self.k = 2
        self._cfg = AdapterConfig(mm_hidden_size=config.mm_hidden_size, hidden_size=config.hidden_size)
        self._packed: Dict = {}

    def _apply(self, fn, *a, **kw):
        self._packed.clear()
        return super()._apply(fn, *a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        self._packed.clear()
        return super()._load_from_state_dict(*a, **kw)

    def packed(self, dtype: torch.dtype) -> ops.PackedGated:
        key = (dtype, str(self.w_gate.device))
        if key not in self._packed:
            sd = {k: v for k, v in self.state_dict().items()}
            self._packed[key] = ops.pack_gated(sd, self._cfg, dtype, self.w_gate.device)
        return self._packed[key]

    @torch.no_grad()
    def forward(self, x, text_embedding=None, attn_mask=None, out_dtype: Optional[torch.dtype] = None,
                operand_dtype: Optional[torch.dtype] = None):
        T = self.target_sequence_length
        if x.shape[0] != T and x.shape[1] != T:                  # compressed local crops: plain MLP
            return self.projection(x, out_dtype=out_dtype, operand_dtype=operand_dtype)
        squeeze = x.dim() <= 2
        if squeeze:
            x = x.unsqueeze(0)
        if x.shape[1] != T:
            raise ValueError(f"GatedBlock expects [N, {T}, D], got {tuple(x.shape)}")
        pg = self.packed(operand_dtype or _operand_dtype(x, self.projection[0].weight))
        out = ops.gated_forward(pg, x, int(self.learnable_gated)).to(out_dtype or x.dtype)
        return out.squeeze(0) if squeeze else out


def build_vision_projector(config, delay_load=False, **kwargs):
    projector_type = getattr(config, "mm_projector_type", "linear")
    if projector_type == "linear":
        return HipLinear(config.mm_hidden_size, config.hidden_size)
    if projector_type == "qformer":
        return Resampler(grid_size=24, embed_dim=config.mm_hidden_size, num_heads=config.mm_hidden_size // 128,
                         kv_dim=config.mm_hidden_size, llm_hidden_size=config.hidden_size)
    if projector_type == "qformer_text":
        raise NotImplementedError("mm_projector_type='qformer_text' (ResamplerWithText) is outside the SliME hot path")
    if projector_type == "gated":
        return GatedBlock(config)
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if m:
        return HipMlp(config.mm_hidden_size, config.hidden_size, int(m.group(1)))
    if projector_type == "identity":
        return IdentityMap()
    raise ValueError(f"Unknown projector type: {projector_type}")
