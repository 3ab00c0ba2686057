"""``build_vision_sampler`` and ``TextGuidedSampler`` (llava/model/multimodal_resampler/builder.py:222-302).

``TextGuidedSampler`` owns ``post_qformer`` (the 144-query local compression ``Resampler``) and the
text-guided top-p router over the merged local tokens.  The cosine router is implemented
(``mm_resampler_type='cosine'``, the released configuration: scripts/llama/llama3_8b_sft.sh:43); the
'qformer' attention router (:136-174) is outside the hot path and raises.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .sampler import IdentityMap, Resampler
from ... import ops


class TextGuidedSampler(nn.Module):
    def __init__(self, projector_type, config):
        super().__init__()
        if projector_type != "cosine":
            raise NotImplementedError(f"mm_resampler_type={projector_type!r}: only the cosine router is on the SliME hot path")
        self.num_queries = config.mm_resampler_dim
        self.topp = config.mm_resampler_topp
        self.temp = config.mm_resampler_temp
        self.grid_size = int(math.sqrt(self.num_queries))
        self.post_qformer = Resampler(grid_size=self.grid_size, embed_dim=config.mm_hidden_size,
                                      num_heads=config.mm_hidden_size // 128, kv_dim=config.mm_hidden_size,
                                      llm_hidden_size=config.hidden_size)

    @torch.no_grad()
    def forward(self, local_f, text_embedding, attn_mask=None):
        """local_f [T, H], text_embedding [L, H], attn_mask [L] -> kept rows of local_f in ascending
        order (resampler/builder.py:248-281, eval path)."""
        keep = ops.router_topp(local_f, text_embedding, attn_mask, float(self.topp), float(self.temp))
        return local_f[keep]


    @torch.no_grad()
    def select_batched(self, tokens, row_off, n_rows, text_embedding, attn_mask=None):
        """Router for all images of a step at once (ops.router_topp_batched: one launch pair, one D2H): per image the kept
        local-token indices, ascending.  Same per-image arithmetic as ``forward``."""
        return ops.router_topp_batched(tokens, row_off, n_rows, text_embedding, attn_mask, float(self.topp), float(self.temp))


def build_vision_sampler(config, delay_load=False, **kwargs):
    mm_resampler_type = getattr(config, "mm_resampler_type", None)
    if mm_resampler_type == "identity" or mm_resampler_type is None:
        return IdentityMap()
    return TextGuidedSampler(mm_resampler_type, config)
