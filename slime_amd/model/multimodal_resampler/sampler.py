"""``Resampler`` -- SliME's local compression layer / the attention expert of the gated adapter -- as a
HIP-backed module with the parameter layout of llava/model/multimodal_resampler/sampler.py:91-137
(``pos_embed`` fp16, ``query``, ``attn.in_proj_*``, ``attn.out_proj.*``, ``ln_q``, ``ln_kv``, ``ln_post``).

Supported configuration = the one the hot path instantiates (resampler/builder.py:239-245,
projector/builder.py:43-50): ``kv_dim == embed_dim`` (kv_proj Identity) and ``use_post_proj=False``.
The query side (``ln_q(query) + pos_embed`` through the q projection) is input independent and is
folded at pack time; the reference's per-call ``isnan(pos_embed)`` device sync (sampler.py:150) does
not exist here.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from ... import ops
from ...weights import sincos_pos_embed_2d


class IdentityMap(nn.Module):
    def __init__(self, *_, **__):
        super().__init__()

    def forward(self, x, *args, **kwargs):
        return x

    @property
    def config(self):
        return {"mm_resampler_type": "identity"}


class Resampler(nn.Module):
    def __init__(self, grid_size, embed_dim, num_heads, kv_dim=None, llm_hidden_size=4096, norm_layer=None,
                 use_post_proj=False, eps: float = 1e-6):
        super().__init__()
        if kv_dim is not None and kv_dim != embed_dim:
            raise NotImplementedError("Resampler with a kv projection is outside the SliME hot path")
        if use_post_proj:
            raise NotImplementedError("Resampler(use_post_proj=True) is outside the SliME hot path")
        self.num_queries = grid_size ** 2
        self.grid_size = grid_size
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.eps = eps
        self.pos_embed = nn.Parameter(torch.from_numpy(sincos_pos_embed_2d(embed_dim, grid_size)).to(torch.float16),
                                      requires_grad=False)
        self.query = nn.Parameter(torch.zeros(self.num_queries, embed_dim))
        nn.init.trunc_normal_(self.query, std=0.02)
        self.kv_proj = nn.Identity()
        self.attn = nn.MultiheadAttention(embed_dim, num_heads)     # parameter container only
        self.ln_q = nn.LayerNorm(embed_dim, eps=eps)
        self.ln_kv = nn.LayerNorm(embed_dim, eps=eps)
        self.ln_post = nn.LayerNorm(embed_dim, eps=eps)
        self.proj = nn.Identity()
        self.compute_dtype = torch.bfloat16
        self._packed: Dict = {}

    def _apply(self, fn, *a, **kw):
        self._packed.clear()
        return super()._apply(fn, *a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        self._packed.clear()
        return super()._load_from_state_dict(*a, **kw)

    def operand_dtype(self, x: Optional[torch.Tensor] = None) -> torch.dtype:
        if x is not None and x.dtype in (torch.bfloat16, torch.float16):
            return x.dtype
        wd = self.query.dtype
        return wd if wd in (torch.bfloat16, torch.float16) else self.compute_dtype

    def packed(self, n_kv: int, dtype: torch.dtype) -> ops.PackedResampler:
        key = (n_kv, dtype, str(self.query.device))
        if key not in self._packed:
            self._packed[key] = ops.pack_resampler(self.state_dict(), self.embed_dim, self.num_heads, n_kv, dtype,
                                                   self.query.device, self.eps)
        return self._packed[key]

    @torch.no_grad()
    def forward(self, x, tgt_size=(24, 24), text=None, attn_mask=None, out_dtype: Optional[torch.dtype] = None,
                operand_dtype: Optional[torch.dtype] = None):
        """x [n, T, D] (or [T, D], squeezed back like sampler.py:141-145,170) -> [n, grid^2, D] in x.dtype.  ``operand_dtype``
        (extension): the 16-bit MFMA operand type for an fp32 ``x`` (see HipMlp.forward)."""
        squeeze = x.dim() <= 2
        if squeeze:
            x = x.unsqueeze(0)
        T = x.shape[1]
        side = int(math.sqrt(T))
        if side * side != T:
            raise ValueError(f"Resampler needs a square key grid, got {T} tokens")
        out = ops.resampler_forward(self.packed(T, operand_dtype or self.operand_dtype(x)), x)
        out = out.to(out_dtype or x.dtype)
        return out.squeeze() if squeeze else out
