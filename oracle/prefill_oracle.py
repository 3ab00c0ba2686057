"""CPU oracle for SURVEY.md section 8 row f-2 (BASELINE configs 4 / 5): the visual-token splice and the Llama-3 prefill attention.

TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module; the product
(slime_amd/) never does.  Plain fp32 torch / integer Python, no reference or HF modules.  Pinned by vectors generated from
the reference's own ``LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal`` and from HF ``LlamaAttention`` (the module
the reference's monkey patch replaces) in ``oracle/make_golden_prefill.py`` -> ``tests/golden/prefill.npz``
(``tests/test_oracle_golden.py``).

  splice_plan / splice                 llava/model/llava_arch.py:343-459  (prepare_inputs_labels_for_multimodal, after encode_images)
  rope_tables / apply_rope             transformers/models/llama/modeling_llama.py (LlamaRotaryEmbedding, apply_rotary_pos_emb,
                                       rotate_half) as called at llava/train/llama_flash_attn_monkey_patch.py:51-54
  llama_attention_forward              llava/train/llama_flash_attn_monkey_patch.py:16-93: q/k/v projections (:31-45), RoPE (:51-54),
                                       repeat_kv (:65-66), causal attention over the un-padded tokens of every sequence
                                       (flash_attn_unpadded_qkvpacked_func(..., causal=True), :79-89), zero rows at padded positions
                                       (pad_input, :90), o_proj (:92)
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

IGNORE_INDEX = -100          # llava/constants.py:7
IMAGE_TOKEN_INDEX = -200     # llava/constants.py:8

PAD_ROW = -1                 # plan entry: zero row (padding)


def feat_row(k: int) -> int:
    """Plan entry for row k of the concatenated image-feature buffer (token ids are >= 0, padding is -1)."""
    return -2 - k


def splice_plan(input_ids: Sequence[Sequence[int]], attention_mask: Optional[Sequence[Sequence[int]]],
                labels: Optional[Sequence[Sequence[int]]], feat_lens: Sequence[int], max_length: Optional[int] = None,
                padding_side: str = "right"):
    """Integer restatement of llava_arch.py:362-459.  ``feat_lens[j]`` = rows of image feature j (features are consumed in
    order of appearance; a sequence WITHOUT an image token still consumes one feature and appends zero rows of it, :377-385).
    Returns (src [B][max_len], labels [B][max_len], mask [B][max_len], position_ids [B][max_len]) as Python int lists:
    src >= 0 is a token id, ``feat_row(k)`` a row of the concatenated features, ``PAD_ROW`` padding."""
    B = len(input_ids)
    starts = [0]
    for n in feat_lens:
        starts.append(starts[-1] + int(n))
    seqs, labs = [], []
    img = 0
    for b in range(B):
        ids = list(input_ids[b])
        keep = [True] * len(ids) if attention_mask is None else [bool(m) for m in attention_mask[b]]
        lab = [IGNORE_INDEX] * len(ids) if labels is None else list(labels[b])
        ids = [t for t, k in zip(ids, keep) if k]                    # :366-368 drop padding via the mask
        lab = [t for t, k in zip(lab, keep) if k]
        n_img = sum(1 for t in ids if t == IMAGE_TOKEN_INDEX)
        if n_img == 0:                                                # :376-385
            seqs.append(ids)
            labs.append(lab)
            img += 1
            continue
        s, l = [], []
        for t, y in zip(ids, lab):                                    # :387-411
            if t == IMAGE_TOKEN_INDEX:
                n = int(feat_lens[img])
                s.extend(feat_row(starts[img] + r) for r in range(n))
                l.extend([IGNORE_INDEX] * n)
                img += 1
            else:
                s.append(t)
                l.append(y)
        seqs.append(s)
        labs.append(l)
    if max_length is not None:                                        # :420-424
        seqs = [s[:max_length] for s in seqs]
        labs = [l[:max_length] for l in labs]
    max_len = max(len(s) for s in seqs)                               # :427
    src = [[PAD_ROW] * max_len for _ in range(B)]
    lab_out = [[IGNORE_INDEX] * max_len for _ in range(B)]
    mask = [[0] * max_len for _ in range(B)]
    pos = [[0] * max_len for _ in range(B)]
    for b, (s, l) in enumerate(zip(seqs, labs)):                      # :435-455
        n = len(s)
        off = max_len - n if padding_side == "left" else 0
        for i in range(n):
            src[b][off + i] = s[i]
            lab_out[b][off + i] = l[i]
            mask[b][off + i] = 1
            pos[b][off + i] = i
    return src, lab_out, mask, pos


def splice(embed_table: Tensor, feats: List[Tensor], input_ids, attention_mask=None, labels=None,
           max_length: Optional[int] = None, padding_side: str = "right"):
    """new_input_embeds [B, max_len, H] (+ labels / mask / position ids) from the plan: pure gather."""
    ids = input_ids.tolist() if isinstance(input_ids, Tensor) else input_ids
    am = None if attention_mask is None else (attention_mask.tolist() if isinstance(attention_mask, Tensor) else attention_mask)
    lb = None if labels is None else (labels.tolist() if isinstance(labels, Tensor) else labels)
    src, lab, mask, pos = splice_plan(ids, am, lb, [f.shape[0] for f in feats], max_length, padding_side)
    allf = torch.cat([f.reshape(-1, f.shape[-1]) for f in feats], 0) if feats else embed_table[:0]
    B, T, H = len(src), len(src[0]), embed_table.shape[1]
    out = torch.zeros((B, T, H), dtype=embed_table.dtype)
    for b in range(B):
        for t in range(T):
            v = src[b][t]
            if v >= 0:
                out[b, t] = embed_table[v]
            elif v != PAD_ROW:
                out[b, t] = allf[-2 - v].to(out.dtype)
    return out, torch.tensor(lab), torch.tensor(mask), torch.tensor(pos)


# ------------------------------------------------------------------------------------------------ Llama attention
def rope_tables(position_ids: Tensor, head_dim: int, theta: float) -> Tuple[Tensor, Tensor]:
    """cos / sin [.., head_dim] in fp32: inv_freq = theta^(-2i/d), emb = cat(freqs, freqs) (LlamaRotaryEmbedding.forward)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = position_ids.to(torch.float32)[..., None] * inv
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """x [B, heads, S, d]; cos/sin [B, S, d]: x*cos + rotate_half(x)*sin with rotate_half(x) = cat(-x2, x1)."""
    d = x.shape[-1] // 2
    rot = torch.cat([-x[..., d:], x[..., :d]], dim=-1)
    return x * cos[:, None] + rot * sin[:, None]


def llama_attention_forward(hidden: Tensor, wq: Tensor, wk: Tensor, wv: Tensor, wo: Tensor, n_heads: int, n_kv_heads: int,
                            position_ids: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None,
                            theta: float = 500000.0, head_dim: Optional[int] = None) -> Tensor:
    """hidden [B, S, D] fp32 -> [B, S, D].  ``attention_mask`` [B, S] is a key-padding mask (1 = token): attention runs causally
    over the un-padded tokens of each sequence in their own order; padded positions give zero rows before o_proj.
    ``head_dim`` defaults to D / n_heads (LlamaConfig's default); a head-sharded slice of a layer (tests/test_dist_gloo.py) has
    fewer heads than D / head_dim and passes it explicitly."""
    B, S, D = hidden.shape
    dh = head_dim or D // n_heads
    g = n_heads // n_kv_heads
    if position_ids is None:
        position_ids = torch.arange(S)[None].expand(B, S)
    q = (hidden @ wq.t()).view(B, S, n_heads, dh).transpose(1, 2)
    k = (hidden @ wk.t()).view(B, S, n_kv_heads, dh).transpose(1, 2)
    v = (hidden @ wv.t()).view(B, S, n_kv_heads, dh).transpose(1, 2)
    cos, sin = rope_tables(position_ids, dh, theta)
    q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
    k = k.repeat_interleave(g, dim=1)                                  # repeat_kv
    v = v.repeat_interleave(g, dim=1)
    out = torch.zeros((B, S, n_heads * dh), dtype=hidden.dtype)
    for b in range(B):
        idx = torch.arange(S) if attention_mask is None else torch.nonzero(attention_mask[b]).squeeze(1)
        n = idx.numel()
        if n == 0:
            continue
        qb, kb, vb = q[b][:, idx], k[b][:, idx], v[b][:, idx]          # [heads, n, dh]
        s = (qb @ kb.transpose(1, 2)) / math.sqrt(dh)
        causal = torch.ones((n, n), dtype=torch.bool).tril()
        s = s.masked_fill(~causal, float("-inf"))
        o = torch.softmax(s, dim=-1) @ vb                              # [heads, n, dh]
        out[b, idx] = o.transpose(0, 1).reshape(n, n_heads * dh)
    return out @ wo.t()
