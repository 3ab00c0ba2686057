"""CPU ORACLE for the SliME visual-encoding hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product (``slime_amd``) never does: it fails loudly when the HIP library is missing.

What it is: a functional, fp32, plain-``torch``-on-CPU restatement of the reference's algorithm
for the path  pixels -> CLIP-ViT-L/14-336 tower -> local compression (``post_qformer``) -> gated
adapter (``mm_projector``) -> spatial merge -> router -> LLM-ready visual tokens.  It holds no
``nn.Module`` of the reference or of ``transformers``; every function cites the reference
``file:line`` it follows (paths relative to the upstream repo root; "HF" = transformers
``models/clip/modeling_clip.py`` -- third-party, pinned ==4.37.2 by the reference's
``pyproject.toml:17``, 5.15.0 installed in the build container; line numbers are 5.15.0's).

How it is pinned ("parity pinned by generated vectors"): the reference ships no tests and no
golden vectors, so ``oracle/make_golden.py`` imports the *reference itself*
in the build container, loads the same seeded weights into the reference's own modules
(``CLIPVisionTower``, ``build_vision_projector``, ``build_vision_sampler``, ``process_images`` ...) and
writes their outputs to ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks this
restatement against those files on every run (CPU, no GPU needed).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ---------------------------------------------------------------------------------------------
# CLIP vision tower (HF CLIPVisionModel as driven by llava/model/multimodal_encoder/clip_encoder.py)
# ---------------------------------------------------------------------------------------------

def _ln(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def quick_gelu(x: Tensor) -> Tensor:
    """HF activations.QuickGELUActivation: x * sigmoid(1.702 x) (CLIP ``hidden_act='quick_gelu'``)."""
    return x * torch.sigmoid(1.702 * x)


def clip_embeddings(sd: Dict[str, Tensor], pixels: Tensor, patch: int) -> Tensor:
    """HF CLIPVisionEmbeddings.forward (modeling_clip.py:196-218): conv(k=stride=patch, no bias),
    flatten(2).transpose(1,2), prepend class_embedding, add position_embedding[0..S)."""
    w = sd["embeddings.patch_embedding.weight"]
    x = F.conv2d(pixels, w, bias=None, stride=patch)            # [N, D, g, g]
    x = x.flatten(2).transpose(1, 2)                            # [N, g*g, D]
    cls = sd["embeddings.class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1)
    return x + sd["embeddings.position_embedding.weight"][None, : x.shape[1]]


def clip_encoder_layer(sd: Dict[str, Tensor], i: int, h: Tensor, heads: int, eps: float) -> Tensor:
    """HF CLIPEncoderLayer.forward (modeling_clip.py:355-383) with CLIPAttention (:259-336, scale
    head_dim**-0.5, softmax in fp32, no mask for the vision tower) and CLIPMLP (:339-352)."""
    p = f"encoder.layers.{i}."
    N, S, D = h.shape
    dh = D // heads
    x = _ln(h, sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], eps)
    q = F.linear(x, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"])
    k = F.linear(x, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
    v = F.linear(x, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
    q = q.view(N, S, heads, dh).transpose(1, 2)
    k = k.view(N, S, heads, dh).transpose(1, 2)
    v = v.view(N, S, heads, dh).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) * dh ** -0.5, dim=-1)
    ctx = (att @ v).transpose(1, 2).reshape(N, S, D)
    h = h + F.linear(ctx, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
    x = _ln(h, sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], eps)
    x = quick_gelu(F.linear(x, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return h + F.linear(x, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def clip_hidden_states(sd: Dict[str, Tensor], cfg, pixels: Tensor, n_layers: Optional[int] = None) -> List[Tensor]:
    """``output_hidden_states=True`` semantics (clip_encoder.py:55; HF CLIPVisionTransformer.forward
    :629-656, CLIPEncoder.forward :455-486): entry 0 = embeddings after ``pre_layrnorm``; entry i =
    output of encoder layer i.  L+1 entries for L layers."""
    L = cfg.num_hidden_layers if n_layers is None else n_layers
    h = clip_embeddings(sd, pixels.float(), cfg.patch_size)
    h = _ln(h, sd["pre_layrnorm.weight"], sd["pre_layrnorm.bias"], cfg.layer_norm_eps)
    out = [h]
    for i in range(L):
        h = clip_encoder_layer(sd, i, h, cfg.num_attention_heads, cfg.layer_norm_eps)
        out.append(h)
    return out


def tower_forward(sd: Dict[str, Tensor], cfg, pixels: Tensor, select_layer: int = -2,
                  select_feature: str = "patch") -> Tensor:
    """CLIPVisionTower.forward + feature_select (clip_encoder.py:36-58): ``hidden_states[select_layer]``,
    drop the cls token for 'patch'.  Only the layers that feed the selected state are evaluated
    (hidden_states[-2] of a 24-layer tower = output of layer 23; layer 24 and post_layernorm are dead)."""
    L = cfg.num_hidden_layers
    idx = select_layer if select_layer >= 0 else L + 1 + select_layer
    if not 0 <= idx <= L:
        raise ValueError(f"select_layer {select_layer} out of range for {L} layers")
    hs = clip_hidden_states(sd, cfg, pixels, n_layers=idx)[idx]
    if select_feature == "patch":
        return hs[:, 1:]
    if select_feature == "cls_patch":
        return hs
    raise ValueError(f"Unexpected select feature: {select_feature}")   # clip_encoder.py:43


# ---------------------------------------------------------------------------------------------
# Resampler (llava/model/multimodal_resampler/sampler.py:91-173)
# ---------------------------------------------------------------------------------------------

def get_abs_pos(abs_pos: Tensor, tgt_size: Tuple[int, int]) -> Tensor:
    """sampler.py:27-36: bicubic (align_corners=False) resize of a square [s*s, D] table; computed
    in fp32 and cast back to the table's dtype (fp16 in the reference)."""
    src = int(math.sqrt(abs_pos.size(0)))
    dtype = abs_pos.dtype
    return F.interpolate(
        abs_pos.float().reshape(1, src, src, -1).permute(0, 3, 1, 2),
        size=(tgt_size[0], tgt_size[1]), mode="bicubic", align_corners=False,
    ).permute(0, 2, 3, 1).flatten(0, 2).to(dtype=dtype)


def mha_forward(q_in: Tensor, k_in: Tensor, v_in: Tensor, in_w: Tensor, in_b: Tensor,
                out_w: Tensor, out_b: Tensor, heads: int) -> Tensor:
    """torch.nn.MultiheadAttention forward, batch-major here ([B, L, E]); packed in_proj split in
    q/k/v thirds, scale head_dim**-0.5, softmax, out_proj (used seq-first at sampler.py:128,162-165)."""
    E = q_in.shape[-1]
    dh = E // heads
    q = F.linear(q_in, in_w[:E], in_b[:E])
    k = F.linear(k_in, in_w[E:2 * E], in_b[E:2 * E])
    v = F.linear(v_in, in_w[2 * E:], in_b[2 * E:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    q = q.view(B, Lq, heads, dh).transpose(1, 2)
    k = k.view(B, Lk, heads, dh).transpose(1, 2)
    v = v.view(B, Lk, heads, dh).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) * dh ** -0.5, dim=-1)
    ctx = (att @ v).transpose(1, 2).reshape(B, Lq, E)
    return F.linear(ctx, out_w, out_b)


def resampler_forward(sd: Dict[str, Tensor], x: Tensor, heads: int, eps: float = 1e-6) -> Tensor:
    """Resampler.forward (sampler.py:140-170) with kv_proj = proj = Identity (kv_dim == embed_dim,
    use_post_proj=False: resampler/builder.py:239-245, projector/builder.py:43-50).
    x: [n, T, D] (or [T, D] -> squeezed back, :141-145,170).  K gets the query grid's sincos table
    bicubically resized to the key grid (:149,155,164); V gets no position term."""
    squeeze = x.dim() <= 2
    if squeeze:
        x = x.unsqueeze(0)
    x = x.float()
    T = x.shape[1]
    tgt = (24, 24)
    if T != tgt[0] * tgt[1]:
        tgt = (int(math.sqrt(T)), int(math.sqrt(T)))                       # :146-147
    pos_q = sd["pos_embed"]                                                # fp16 table
    pos_k = get_abs_pos(pos_q, tgt)                                        # fp16 again
    xn = _ln(x, sd["ln_kv.weight"].float(), sd["ln_kv.bias"].float(), eps)   # :158
    q = _ln(sd["query"].float(), sd["ln_q.weight"].float(), sd["ln_q.bias"].float(), eps)   # :161
    q_in = (q + pos_q.float()).unsqueeze(0).expand(x.shape[0], -1, -1)     # :163
    out = mha_forward(q_in, xn + pos_k.float().unsqueeze(0), xn,
                      sd["attn.in_proj_weight"].float(), sd["attn.in_proj_bias"].float(),
                      sd["attn.out_proj.weight"].float(), sd["attn.out_proj.bias"].float(), heads)
    out = _ln(out, sd["ln_post.weight"].float(), sd["ln_post.bias"].float(), eps)   # :168
    return out.squeeze(0) if squeeze else out


# ---------------------------------------------------------------------------------------------
# mm_projector (llava/model/multimodal_projector/builder.py)
# ---------------------------------------------------------------------------------------------

def mlp_projector(sd: Dict[str, Tensor], x: Tensor) -> Tensor:
    """``projection`` = Linear -> nn.GELU() (exact erf) -> Linear (projector/builder.py:53-57; the same
    shape as ``mlp2x_gelu`` :241-248)."""
    x = F.linear(x.float(), sd["projection.0.weight"].float(), sd["projection.0.bias"].float())
    x = F.gelu(x)
    return F.linear(x, sd["projection.2.weight"].float(), sd["projection.2.bias"].float())


def gate_weights(sd: Dict[str, Tensor], x2d: Tensor) -> Tensor:
    """noisy_top_k_gating in eval mode (projector/builder.py:137-176): softmax over the 2 experts of
    x @ w_gate, top-2-of-2, renormalised by (sum + 1e-6), scattered back -> [T, 2] in expert order."""
    logits = x2d.float() @ sd["w_gate"].float()
    p = torch.softmax(logits, dim=1)
    top, idx = p.topk(2, dim=1)
    gates = top / (top.sum(1, keepdim=True) + 1e-6)
    return torch.zeros_like(p).scatter(1, idx, gates)


def gated_block_forward(sd: Dict[str, Tensor], x: Tensor, heads: int, learnable_gated: int = -1,
                        target_len: int = 576) -> Tensor:
    """GatedBlock.forward (projector/builder.py:179-209).  Inputs whose dim0 and dim1 are both
    != 576 take the early return ``projection(x)`` (:180-181) -- what the compressed local crops
    [n,144,D] hit.  Otherwise E0 = projection(x), E1 = projection(attn(x)); ``mm_learnable_gated>=0``
    returns that expert (:198-201); else the softmax-gated mix (:203-206)."""
    if x.shape[0] != target_len and x.shape[1] != target_len:
        return mlp_projector(sd, x)
    squeeze = x.dim() <= 2
    if squeeze:
        x = x.unsqueeze(0)
    x = x.float()
    attn_sd = {k[len("attn."):]: v for k, v in sd.items() if k.startswith("attn.")}
    e0 = mlp_projector(sd, x)
    e1 = mlp_projector(sd, resampler_forward(attn_sd, x, heads))
    if learnable_gated >= 0:
        out = (e0, e1)[learnable_gated]
    else:
        N, C, D = x.shape
        g = gate_weights(sd, x.reshape(N * C, D)).reshape(N, C, 2)
        out = e0 * g[..., 0:1] + e1 * g[..., 1:2]
    return out.squeeze(0) if squeeze else out


# ---------------------------------------------------------------------------------------------
# Slicer grid logic (llava/mm_utils.py) -- integer/float-ratio arithmetic, needed by the merge
# ---------------------------------------------------------------------------------------------

def _factor_pairs(n: int) -> List[Tuple[int, int]]:
    return [(i, n // i) for i in range(1, n + 1) if n % i == 0]


def select_best_resolution_uhd(original_size: Tuple[int, int], processor_size=(336, 336)) -> Tuple[int, int]:
    """mm_utils.py:41-97.  scale = ceil(W*H / 336^2), capped at 6, and 1 is bumped to 2 (:56-59);
    candidates are the (w,h) factor pairs of {scale, scale+1} if scale <= 2 else
    {scale-1, scale, scale+1} (:78-81); keep the max effective resolution, tie -> min waste,
    first wins (:86-96)."""
    iw, ih = processor_size
    ow, oh = original_size
    scale = math.ceil(ow * oh / (iw * ih))
    if scale > 6:
        scale = 6
    elif scale == 1:
        scale = 2
    cands = (_factor_pairs(scale) + _factor_pairs(scale + 1)) if scale <= 2 else \
        (_factor_pairs(scale - 1) + _factor_pairs(scale) + _factor_pairs(scale + 1))
    best, max_eff, min_waste = None, 0, float("inf")
    for ws, hs in cands:
        width, height = ws * iw, hs * ih
        s = min(width / ow, height / oh)
        dw, dh = int(ow * s), int(oh * s)
        eff = min(dw * dh, ow * oh)
        waste = width * height - eff
        if eff > max_eff or (eff == max_eff and waste < min_waste):
            max_eff, min_waste, best = eff, waste, (width, height)
    return best


def anyres_grid_shape(image_size: Tuple[int, int], patch_size: int = 336) -> Tuple[int, int]:
    """get_anyres_image_grid_shape (mm_utils.py:156-174): the pinpoint result is overwritten by the
    uhd result for a hard-coded (336,336) (:172-173), so ``grid_pinpoints`` is dead.  -> (nw, nh)."""
    w, h = select_best_resolution_uhd(image_size, (336, 336))
    return w // patch_size, h // patch_size


# ---------------------------------------------------------------------------------------------
# encode_images glue (llava/model/llava_arch.py:212-255, sampler branch)
# ---------------------------------------------------------------------------------------------

def spatial_merge(local: Tensor, nw: int, nh: int, grid: int) -> Tensor:
    """llava_arch.py:235-244: [n, grid*grid, C] -> view(nh, nw, g, g, C).permute(0,2,1,3,4)
    .flatten(0,3): whole-image raster order of the compressed local tokens."""
    C = local.shape[-1]
    return local.reshape(nh, nw, grid, grid, C).permute(0, 2, 1, 3, 4).reshape(nh * grid * nw * grid, C)


def router_cosine_scores(image: Tensor, text: Tensor, attn_mask: Optional[Tensor]) -> Tensor:
    """TextGuidedRouterCosine.forward (resampler/builder.py:177-201): cosine similarity of every
    local token with every text embedding; masked positions zeroed then summed (mean if no mask)."""
    sim = F.cosine_similarity(image.float().unsqueeze(1), text.float().unsqueeze(0), dim=-1)
    if attn_mask is not None:
        sim = sim.masked_fill((attn_mask == False).unsqueeze(0), 0.0)   # noqa: E712 (as the reference)
        return sim.sum(dim=-1)
    return sim.mean(dim=-1)


def router_select(scores: Tensor, topp: float, temp: float) -> Tensor:
    """TextGuidedSampler.forward eval path (resampler/builder.py:248-281): softmax(scores/temp),
    sort descending, cumsum, keep the prefix with cumsum <= topp plus one more element (:266-270),
    return the kept token indices in ascending order."""
    probs = torch.softmax(scores / temp, dim=-1)
    sorted_probs, sorted_idx = torch.sort(probs, descending=True)
    cum = torch.cumsum(sorted_probs, dim=0)
    sel = (cum <= topp).nonzero(as_tuple=True)[0]
    if sel.numel() < sorted_idx.numel():
        sel = sorted_idx[: sel.numel() + 1]
    return sel.sort(descending=False)[0]


def encode_image(tower_sd, adapter_sd, vcfg, acfg, crops: Tensor, image_size: Tuple[int, int],
                 text_emb: Optional[Tensor] = None, text_mask: Optional[Tensor] = None,
                 separator: Optional[Tensor] = None, merge: str = "spatial",
                 topp: float = 0.95, temp: float = 1.0, select_layer: int = -2) -> Dict[str, Tensor]:
    """One image of encode_images' sampler branch (llava_arch.py:217-255).  ``crops`` [1+n,3,336,336]:
    crop 0 is the global view.  Returns every intermediate stage so fixtures can pin each one."""
    from slime_amd.weights import sub_state   # key helper only; no product compute
    proj_sd = sub_state(adapter_sd, "mm_projector.")
    post_sd = sub_state(adapter_sd, "sampler.post_qformer.")
    feats = tower_forward(tower_sd, vcfg, crops, select_layer)                   # :222
    g = gated_block_forward(proj_sd, feats[0], acfg.num_heads)                   # :224
    comp = resampler_forward(post_sd, feats[1:], acfg.num_heads, acfg.ln_eps)    # :226
    loc = gated_block_forward(proj_sd, comp, acfg.num_heads)                     # :227 (early return)
    grid = int(math.isqrt(acfg.local_queries))
    if merge == "flat":
        merged = loc.flatten(0, 1)                                               # :233-234
    else:
        nw, nh = anyres_grid_shape(image_size)
        merged = spatial_merge(loc, nw, nh, grid)                                # :235-244
    out = {"tower": feats, "global": g, "compressed": comp, "local": loc, "merged": merged}
    if text_emb is not None:
        scores = router_cosine_scores(merged, text_emb, text_mask)               # :248
        keep = router_select(scores, topp, temp)
        out["router_scores"], out["router_keep"] = scores, keep
        merged = merged[keep]
    if separator is not None:
        out["tokens"] = torch.cat([g, separator.float().reshape(1, -1), merged], dim=0)   # :254-255
    return out
