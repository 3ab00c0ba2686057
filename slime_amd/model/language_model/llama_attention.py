"""Llama-3 prefill self-attention on the HIP path: the counterpart of llava/train/llama_flash_attn_monkey_patch.py.

The reference swaps ``LlamaAttention.forward`` for a flash-attn version (``replace_llama_attn_with_flash_attn``, :105-115);
``replace_llama_attn_with_hip_attn`` does the same with ``slime_llama_attn_forward`` (fused q/k/v projection GEMM, RoPE,
causal grouped-query attention over the un-padded tokens, o_proj GEMM -- slime_amd/csrc/prefill.hip).  ``HipLlamaAttention``
is the standalone module form (HF parameter names ``q_proj`` / ``k_proj`` / ``v_proj`` / ``o_proj``, no biases) used by the
tests and bench.py.  Prefill only: a KV cache (``past_key_value``) is outside this row of SURVEY.md section 8 and raises.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from ... import ops


class HipLlamaAttention(nn.Module):
    def __init__(self, hidden_size: int = 4096, num_heads: int = 32, num_key_value_heads: int = 8, head_dim: int = 128,
                 rope_theta: float = 500000.0, compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        if head_dim != 128:
            raise NotImplementedError("the HIP prefill attention is built for head_dim 128 (Llama-3)")
        self.hidden_size, self.num_heads, self.num_key_value_heads, self.head_dim = hidden_size, num_heads, num_key_value_heads, head_dim
        self.num_key_value_groups = num_heads // num_key_value_heads
        self.rope_theta = rope_theta
        self.compute_dtype = compute_dtype
        self.q_proj = nn.Linear(hidden_size, num_heads * head_dim, bias=False)
        self.k_proj = nn.Linear(hidden_size, num_key_value_heads * head_dim, bias=False)
        self.v_proj = nn.Linear(hidden_size, num_key_value_heads * head_dim, bias=False)
        self.o_proj = nn.Linear(num_heads * head_dim, hidden_size, bias=False)
        self._packed: Dict = {}

    def _apply(self, fn, *a, **kw):
        self._packed.clear()
        return super()._apply(fn, *a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        self._packed.clear()
        return super()._load_from_state_dict(*a, **kw)

    def packed(self, dtype: torch.dtype) -> ops.PackedLlamaAttention:
        key = (dtype, str(self.q_proj.weight.device))
        if key not in self._packed:
            self._packed[key] = ops.pack_llama_attention(self.q_proj.weight, self.k_proj.weight, self.v_proj.weight,
                                                         self.o_proj.weight, self.num_heads, self.num_key_value_heads, dtype,
                                                         self.q_proj.weight.device, self.rope_theta)
        return self._packed[key]

    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.Tensor] = None, past_key_value=None, output_attentions: bool = False,
                use_cache: bool = False, **_):
        """Signature and return triple of the patched forward (llama_flash_attn_monkey_patch.py:16-24,92):
        ``attention_mask`` is the [B, S] key-padding mask (the patch disables HF's 4-D mask expansion, :97-102)."""
        if past_key_value is not None or use_cache:
            raise NotImplementedError("HipLlamaAttention implements the prefill pass (no KV cache)")
        dt = hidden_states.dtype if hidden_states.dtype in (torch.bfloat16, torch.float16) else self.compute_dtype
        out = ops.llama_attention_forward(self.packed(dt), hidden_states, position_ids, attention_mask, hidden_states.dtype)
        return out, None, None


def _rope_theta(cfg) -> float:
    """rope_theta of a LlamaConfig across transformers versions (4.x attribute, 5.x ``rope_parameters`` dict); only the default
    rotary embedding is implemented in the kernel (``slime_rope``: inv_freq = theta^(-2i/d), as Llama-3-8B / SliME-8B use)."""
    rp = getattr(cfg, "rope_parameters", None)
    scaling = getattr(cfg, "rope_scaling", None) or (rp if isinstance(rp, dict) else None)
    if isinstance(scaling, dict):
        kind = scaling.get("rope_type", scaling.get("type", "default"))
        if kind not in (None, "default"):
            raise NotImplementedError(f"slime_amd's prefill attention implements the default rotary embedding, not rope_type={kind!r}")
    if isinstance(rp, dict) and rp.get("rope_theta") is not None:
        return float(rp["rope_theta"])
    return float(getattr(cfg, "rope_theta", 500000.0))


def _self_attn_return_arity(M) -> int:
    """How many values ``LlamaDecoderLayer.forward`` unpacks from ``self.self_attn(...)``: 3 in transformers <= 4.47 (attn output,
    weights, present key/value -- the reference's pin 4.37.2, llama_flash_attn_monkey_patch.py:92), 2 from 4.48 on."""
    import inspect
    import re
    try:
        src = inspect.getsource(M.LlamaDecoderLayer.forward)
        # the assignment targets on the ONE line that calls self.self_attn( -- anchored so that nothing above it can join the match
        m = re.search(r"^[ \t]*([^\n=]+?)=[ \t]*self\.self_attn\(", src, re.M)
        if m:
            return len([t for t in m.group(1).split(",") if t.strip()])
    except (OSError, TypeError):
        pass
    import transformers
    major, minor = (int(x) for x in transformers.__version__.split(".")[:2])
    return 2 if (major, minor) >= (4, 48) else 3


_PATCH_STATE: Dict[str, object] = {}


def replace_llama_attn_with_hip_attn():
    """Patch HF ``LlamaAttention.forward`` like ``replace_llama_attn_with_flash_attn`` does (llama_flash_attn_monkey_patch.py:105-115),
    for whichever transformers is installed:

      * the forward accepts the keyword set of every generation (``position_ids``, ``past_key_value`` / ``past_key_values``,
        ``position_embeddings``, ``cache_position`` ...) and returns as many values as the installed ``LlamaDecoderLayer`` unpacks
        (3 up to 4.47, 2 from 4.48 on);
      * the [B, S] key-padding mask must reach the attention un-expanded, as in the reference's ``_prepare_decoder_attention_mask``
        override (:97-102): 4.37 keeps that hook; 4.38-4.52 build the 4-D mask in ``LlamaModel._update_causal_mask``; later
        versions call the module-level ``create_causal_mask`` -- each is replaced by a pass-through of the 2-D mask.

    Returns the ``transformers`` module; ``restore_llama_attn()`` undoes the patch."""
    import transformers
    from transformers.models.llama import modeling_llama as M
    arity = _self_attn_return_arity(M)
    import inspect
    stock = _PATCH_STATE.get("forward", M.LlamaAttention.forward)
    # positional parameters of the INSTALLED LlamaAttention.forward after hidden_states: (attention_mask, position_ids, past_key_value,
    # ...) up to 4.47, (position_embeddings, attention_mask, past_key_value(s), cache_position) from 4.48 on -- extras are mapped by name
    positional = [n for n, q in list(inspect.signature(stock).parameters.items())[2:]
                  if q.kind in (q.POSITIONAL_ONLY, q.POSITIONAL_OR_KEYWORD)]

    def forward(self, hidden_states, *args, **kw):
        if len(args) > len(positional):
            raise TypeError(f"LlamaAttention.forward takes at most {len(positional) + 1} positional arguments ({len(args) + 1} given)")
        for name, value in zip(positional, args):
            if name in kw:
                raise TypeError(f"LlamaAttention.forward got multiple values for argument {name!r}")
            kw[name] = value
        attention_mask, position_ids = kw.get("attention_mask"), kw.get("position_ids")
        past_key_value, past_key_values = kw.get("past_key_value"), kw.get("past_key_values")
        output_attentions, use_cache = kw.get("output_attentions", False), kw.get("use_cache", False)
        # prefill only: HF generate() defaults to use_cache=True -- call the patched model with use_cache=False (a present
        # k/v return for the prefill step would be the natural extension; decode is outside SURVEY section 8)
        if past_key_value is not None or past_key_values is not None or use_cache:
            raise NotImplementedError("slime_amd patches the prefill pass only: pass use_cache=False (no KV cache on this path)")
        if output_attentions:
            raise NotImplementedError("the fused prefill attention never materialises the attention weights")
        cfg = self.config
        key = "_slime_packed"
        dt = hidden_states.dtype if hidden_states.dtype in (torch.bfloat16, torch.float16) else torch.bfloat16
        cache = self.__dict__.setdefault(key, {})
        # the packed copy is keyed on the identity AND the in-place version of the four weights: load_state_dict, a LoRA merge or
        # any other in-place edit after the first forward bumps ``_version`` and re-packs instead of silently using stale copies
        ws = (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.o_proj.weight)
        k = (dt, str(ws[0].device)) + tuple((w.data_ptr(), w._version) for w in ws)
        if k not in cache:
            cache.clear()
            cache[k] = ops.pack_llama_attention(self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.o_proj.weight,
                                                cfg.num_attention_heads, cfg.num_key_value_heads, dt, self.q_proj.weight.device,
                                                _rope_theta(cfg))
        if attention_mask is not None and attention_mask.dim() != 2:
            raise ValueError("expected the [B, S] key-padding mask (llama_flash_attn_monkey_patch.py:97-102): the model-level mask "
                             "builder of this transformers version is not patched through")
        out = ops.llama_attention_forward(cache[k], hidden_states, position_ids, attention_mask, hidden_states.dtype)
        return (out, None) if arity == 2 else (out, None, None)

    if not _PATCH_STATE:
        _PATCH_STATE["forward"] = M.LlamaAttention.forward
        for name in ("_prepare_decoder_attention_mask", "_update_causal_mask"):
            if hasattr(M.LlamaModel, name):
                _PATCH_STATE["model." + name] = getattr(M.LlamaModel, name)
        if hasattr(M, "create_causal_mask"):
            _PATCH_STATE["module.create_causal_mask"] = M.create_causal_mask
    if hasattr(M.LlamaModel, "_prepare_decoder_attention_mask"):                 # <= 4.37: the reference's own hook
        M.LlamaModel._prepare_decoder_attention_mask = lambda self, attention_mask, *a, **k: attention_mask
    if hasattr(M.LlamaModel, "_update_causal_mask"):                             # 4.38 ... 4.52
        M.LlamaModel._update_causal_mask = lambda self, attention_mask, *a, **k: attention_mask
    if hasattr(M, "create_causal_mask"):                                         # >= 4.53: LlamaModel.forward looks the name up in its module
        M.create_causal_mask = lambda *a, attention_mask=None, **k: attention_mask
    M.LlamaAttention.forward = forward
    return transformers


def restore_llama_attn():
    """Undo ``replace_llama_attn_with_hip_attn`` (tests run patched and stock models in one process)."""
    if not _PATCH_STATE:
        return
    from transformers.models.llama import modeling_llama as M
    M.LlamaAttention.forward = _PATCH_STATE["forward"]
    for k, v in _PATCH_STATE.items():
        if k.startswith("model."):
            setattr(M.LlamaModel, k[6:], v)
        elif k.startswith("module."):
            setattr(M, k[7:], v)
    _PATCH_STATE.clear()
