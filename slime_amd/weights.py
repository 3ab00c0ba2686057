"""Deterministic synthetic weights + state-dict key layout for the SliME visual hot path.

There is no network on either box, so neither the CLIP-ViT-L/14-336 checkpoint nor a SliME
checkpoint is available.  Both the build container (where the reference is imported to make
golden vectors) and the GPU box regenerate *identical* weights from a seed with a CPU
``torch.Generator`` (bit-reproducible for a fixed torch build).

Key layout follows the reference's checkpoints:

* tower  : HF ``CLIPVisionModel`` keys.  transformers==4.37.2 (pinned by the reference,
  ``pyproject.toml:17``) prefixes them with ``vision_model.``; transformers 5.x dropped the
  prefix.  :func:`canonical_tower_key` accepts both.
* adapter: ``mm_projector.*`` (GatedBlock, ``llava/model/multimodal_projector/builder.py:38-74``) and
  ``sampler.post_qformer.*`` (Resampler, ``llava/model/multimodal_resampler/sampler.py:115-137``).

The scales differ on purpose from HF's ``_init_weights``: q/k projections are O(1) so that the
attention logits have unit-order spread (HF's init makes softmax almost uniform, which would
make attention parity vacuous); biases and LayerNorm affine parameters are non-trivial so every
epilogue path is exercised; ``w_gate`` is non-zero (the reference zero-inits it, which makes the
gate a constant 0.5/0.5).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict
from typing import Dict

import numpy as np
import torch


@dataclass(frozen=True)
class VisionConfig:
    """Geometry of the CLIP vision tower (HF ``CLIPVisionConfig`` field names)."""
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 336
    patch_size: int = 14
    layer_norm_eps: float = 1e-5
    hidden_act: str = "quick_gelu"

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def seq_len(self) -> int:
        return self.num_patches + 1

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def to_dict(self):
        return asdict(self)


CLIP_L_336 = VisionConfig()
# Small geometry used by the golden-vector fixtures: full tensors fit in a few hundred KB.
# Same image/patch geometry (336/14 -> S=577) so every shape-dependent code path is shared.
TINY = VisionConfig(hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2)


@dataclass(frozen=True)
class AdapterConfig:
    """Geometry of the SliME adapter (``mm_projector_type='gated'`` + ``post_qformer``)."""
    mm_hidden_size: int = 1024      # tower width
    hidden_size: int = 4096         # LLM width
    head_dim: int = 128             # resamplers use embed_dim // 128 heads (projector/builder.py:46)
    global_queries: int = 576       # GatedBlock.attn grid 24x24 (projector/builder.py:41-43)
    local_queries: int = 144        # mm_resampler_dim (scripts/llama/llama3_8b_sft.sh:46)
    ln_eps: float = 1e-6            # Resampler norm_layer eps (sampler.py:106)

    @property
    def num_heads(self) -> int:
        return self.mm_hidden_size // self.head_dim


ADAPTER_8B = AdapterConfig()
ADAPTER_TINY = AdapterConfig(mm_hidden_size=128, hidden_size=256)   # 1 head of 128 (reference: width // 128)


def _gen(seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def _normal(g, shape, std, mean=0.0):
    return torch.randn(shape, generator=g, dtype=torch.float32) * std + mean


def make_tower_state_dict(cfg: VisionConfig = CLIP_L_336, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """fp32 state dict with transformers-4.37 key names (``vision_model.`` prefix)."""
    g = _gen(seed)
    D, F, P = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size
    sd: Dict[str, torch.Tensor] = {}
    p = "vision_model."
    sd[p + "embeddings.class_embedding"] = _normal(g, (D,), D ** -0.5)
    sd[p + "embeddings.patch_embedding.weight"] = _normal(g, (D, 3, P, P), 0.02)
    sd[p + "embeddings.position_embedding.weight"] = _normal(g, (cfg.seq_len, D), 0.02)
    sd[p + "pre_layrnorm.weight"] = _normal(g, (D,), 0.1, 1.0)   # sic: HF's typo is the real key
    sd[p + "pre_layrnorm.bias"] = _normal(g, (D,), 0.05)
    depth_scale = (2 * cfg.num_hidden_layers) ** -0.5
    for i in range(cfg.num_hidden_layers):
        q = f"{p}encoder.layers.{i}."
        sd[q + "layer_norm1.weight"] = _normal(g, (D,), 0.1, 1.0)
        sd[q + "layer_norm1.bias"] = _normal(g, (D,), 0.05)
        for name, std in (("q_proj", D ** -0.5), ("k_proj", D ** -0.5),
                          ("v_proj", D ** -0.5), ("out_proj", D ** -0.5 * depth_scale * 2.0)):
            sd[q + f"self_attn.{name}.weight"] = _normal(g, (D, D), std)
            sd[q + f"self_attn.{name}.bias"] = _normal(g, (D,), 0.02)
        sd[q + "layer_norm2.weight"] = _normal(g, (D,), 0.1, 1.0)
        sd[q + "layer_norm2.bias"] = _normal(g, (D,), 0.05)
        sd[q + "mlp.fc1.weight"] = _normal(g, (F, D), (2 * D) ** -0.5 * 2.0)
        sd[q + "mlp.fc1.bias"] = _normal(g, (F,), 0.02)
        sd[q + "mlp.fc2.weight"] = _normal(g, (D, F), F ** -0.5 * depth_scale * 2.0)
        sd[q + "mlp.fc2.bias"] = _normal(g, (D,), 0.02)
    sd[p + "post_layernorm.weight"] = _normal(g, (D,), 0.1, 1.0)
    sd[p + "post_layernorm.bias"] = _normal(g, (D,), 0.05)
    return sd


def canonical_tower_key(key: str) -> str:
    """Map either HF naming generation onto the prefix-free form used internally."""
    return key[len("vision_model."):] if key.startswith("vision_model.") else key


def strip_tower_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {canonical_tower_key(k): v for k, v in sd.items()}


# --- 2-D sin/cos position table of the Resampler (sampler.py:39-88), restated with numpy -----------

def sincos_pos_embed_2d(embed_dim: int, grid_size: int) -> np.ndarray:
    """[grid*grid, embed_dim] float64->float32 table; first half encodes the w index, second the h
    index (``np.meshgrid(grid_w, grid_h)`` puts w first, sampler.py:45-49)."""
    assert embed_dim % 4 == 0
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)

    def one_d(dim, pos):
        omega = np.arange(dim // 2, dtype=np.float32)
        omega /= dim / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    return np.concatenate([one_d(embed_dim // 2, grid[0]), one_d(embed_dim // 2, grid[1])], axis=1)


def _resampler_state(g, prefix: str, D: int, n_query_side: int) -> Dict[str, torch.Tensor]:
    nq = n_query_side * n_query_side
    sd = {}
    # pos_embed is an fp16 Parameter in the reference (sampler.py:115-117)
    sd[prefix + "pos_embed"] = torch.from_numpy(sincos_pos_embed_2d(D, n_query_side)).to(torch.float16)
    sd[prefix + "query"] = _normal(g, (nq, D), 0.5)
    sd[prefix + "attn.in_proj_weight"] = _normal(g, (3 * D, D), D ** -0.5)
    sd[prefix + "attn.in_proj_bias"] = _normal(g, (3 * D,), 0.02)
    sd[prefix + "attn.out_proj.weight"] = _normal(g, (D, D), D ** -0.5)
    sd[prefix + "attn.out_proj.bias"] = _normal(g, (D,), 0.02)
    for ln in ("ln_q", "ln_kv", "ln_post"):
        sd[prefix + ln + ".weight"] = _normal(g, (D,), 0.1, 1.0)
        sd[prefix + ln + ".bias"] = _normal(g, (D,), 0.05)
    return sd


def make_adapter_state_dict(cfg: AdapterConfig = ADAPTER_8B, seed: int = 4321) -> Dict[str, torch.Tensor]:
    """State dict for ``mm_projector`` (GatedBlock) and ``sampler`` (post_qformer), reference key names."""
    g = _gen(seed)
    D, H = cfg.mm_hidden_size, cfg.hidden_size
    sd: Dict[str, torch.Tensor] = {}
    sd["mm_projector.projection.0.weight"] = _normal(g, (H, D), D ** -0.5)
    sd["mm_projector.projection.0.bias"] = _normal(g, (H,), 0.02)
    sd["mm_projector.projection.2.weight"] = _normal(g, (H, H), H ** -0.5)
    sd["mm_projector.projection.2.bias"] = _normal(g, (H,), 0.02)
    sd.update(_resampler_state(g, "mm_projector.attn.", D, int(math.isqrt(cfg.global_queries))))
    # bf16 Parameters in the reference (projector/builder.py:63-64); w_noise is unused at inference
    sd["mm_projector.w_gate"] = _normal(g, (D, 2), 0.05).to(torch.bfloat16)
    sd["mm_projector.w_noise"] = torch.zeros(D, 2, dtype=torch.bfloat16)
    sd["mm_projector.mean"] = torch.tensor([0.0], dtype=torch.bfloat16)
    sd["mm_projector.std"] = torch.tensor([1.0], dtype=torch.bfloat16)
    sd.update(_resampler_state(g, "sampler.post_qformer.", D, int(math.isqrt(cfg.local_queries))))
    return sd


def sub_state(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    """Entries under ``prefix`` with the prefix removed."""
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def synthetic_pixels(n_crops: int, seed: int = 0, image_size: int = 336) -> torch.Tensor:
    """N(0,1) ``pixel_values`` [n,3,S,S] fp32 (synthetic inputs)."""
    g = _gen(seed)
    return torch.randn((n_crops, 3, image_size, image_size), generator=g, dtype=torch.float32)
