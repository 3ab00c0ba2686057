"""Torch-tensor front end of libslime_hip: weight packing (load time) and thin call wrappers.

PyTorch is plumbing here -- device memory, streams, dtype bookkeeping.  Every arithmetic step of the
hot path runs in the HIP library; there is no eager / CPU fallback (``_lib.load()`` raises if the
library is absent).
"""
from __future__ import annotations

import collections
import ctypes as C
import math
import os
import threading
import weakref
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import _lib
from .weights import VisionConfig, AdapterConfig, strip_tower_prefix, sub_state

LOG2E = 1.4426950408889634   # slime_attention takes q pre-scaled by head_dim^-0.5 * log2(e)

_DT = {torch.bfloat16: _lib.BF16, torch.float16: _lib.F16, torch.float32: _lib.F32, torch.uint8: _lib.U8}


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise ValueError(f"unsupported dtype {dt}") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "libslime_hip needs contiguous device tensors"
    return t.data_ptr()


def _require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise _lib.SlimeHipError(f"{name} must live on the GPU: slime_amd has no CPU path")


class Workspace:
    """Grow-only scratch buffer (256-B aligned by the caching allocator's 512-B granularity)."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != torch.device(device):
            self.buf = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
        return self.buf


# ------------------------------------------------------------------------------------------------
# primitive wrappers (used by the parity tests and by the modules)
# ------------------------------------------------------------------------------------------------

def pack_b_frag(w: torch.Tensor) -> Optional[torch.Tensor]:
    """Copy of the static GEMM operand(s) w T [..., N, K] in MFMA-fragment order (slime_gemm_pack_b; layout in
    include/slime_hip.h), or None where the library cannot run from it (it is asked: slime_gemm_b_frag_usable; host tensors)."""
    if w is None or not w.is_cuda:
        return None
    N, K = w.shape[-2], w.shape[-1]
    lib = _lib.load()
    if not lib.slime_gemm_b_frag_usable(N, K):
        return None
    w = w.contiguous()
    out = torch.empty_like(w)
    per = N * K * w.element_size()
    assert lib.slime_gemm_packed_b_bytes(N, K) == per
    for i in range(w.numel() // (N * K)):
        _lib.check(lib.slime_gemm_pack_b(w.data_ptr() + i * per, N, K, out.data_ptr() + i * per, _stream()), "slime_gemm_pack_b")
    return out


def keep_row_major() -> bool:
    """Weight-memory policy.  Since ABI 5 every kernel slime_gemm_ex dispatches to can read a static operand from
    its fragment-order image alone, so by default a packed weight is resident ONCE (tower 0.58 GB, adapter 62 MB; through round 4
    every weight was held twice: row-major for the LDS-staged kernels + fragment order for the direct-B kernel).
    SLIME_KEEP_ROW_MAJOR=1 keeps the row-major copies as well (the round-4 layout; results are bit-identical either way)."""
    return os.environ.get("SLIME_KEEP_ROW_MAJOR", "0") == "1"


def pack_static(T: Dict[str, Optional[torch.Tensor]], names: Sequence[str]) -> None:
    """T[name + '_frag'] = fragment-order image of T[name]; the row-major tensor is dropped (T[name] = None) where the library can
    run from the image alone, unless keep_row_major()."""
    for name in names:
        T[name + "_frag"] = pack_b_frag(T[name])
        if T[name + "_frag"] is not None and not keep_row_major():
            T[name] = None


def packed_weight_bytes(*packs) -> int:
    """Resident bytes of the packed tensors of PackedTower / PackedResampler / PackedMlp / ... objects (each tensor once)."""
    seen, total = set(), 0
    for p in packs:
        for t in (p.tensors if hasattr(p, "tensors") else p).values():
            if isinstance(t, torch.Tensor) and t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                total += t.numel() * t.element_size()
    return total


def gemm(a: torch.Tensor, w: Optional[torch.Tensor], bias: Optional[torch.Tensor], epilogue: int,
         out: Optional[torch.Tensor] = None, w_frag: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None,
         row_map: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = epi(a @ w.T + bias); a [M,K] T, w [N,K] T, bias fp32 [N]; w_frag = pack_b_frag(w) (optional); resid T [M,N] for
    EPI_BIAS_RESID_T (out = T(a @ w.T + bias + resid), out may be resid); row_map int32 [M] (device): row r of the result is stored
    at out[row_map[r]] (``out`` given, with at least max(row_map) + 1 rows; plain T / fp32 epilogues)."""
    lib = _lib.load()
    M, K = a.shape
    if w is None and w_frag is None:
        raise ValueError("gemm: the static operand is missing (row-major w, its fragment-order image w_frag, or both)")
    N = (w if w is not None else w_frag).shape[0]        # w = None: the fragment-order image alone (slime_gemm_b_frag_usable)
    if out is None:
        odt = a.dtype if (epilogue <= _lib.EPI_BIAS_GELU_T or epilogue == _lib.EPI_BIAS_RESID_T) else torch.float32
        out = torch.empty((M, N), dtype=odt, device=a.device)
    g = _lib.GemmArgs(A=_ptr(a), lda=a.stride(0), B=_ptr(w), bias=_ptr(bias), C=_ptr(out), ldc=out.stride(0), M=M, N=N, K=K,
                      dtype=dtype_code(a.dtype), epilogue=epilogue, B_frag=_ptr(w_frag), resid=_ptr(resid),
                      ldr=resid.stride(0) if resid is not None else 0, row_map=_ptr(row_map))
    _lib.check(lib.slime_gemm_ex(C.byref(g), _stream()), "slime_gemm_ex")
    return out


def gemm_ln_producer(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], h: torch.Tensor,
                     w_frag: Optional[torch.Tensor] = None):
    """h (fp32, in place) += a @ w.T + bias; returns (x16 = T(h), stats [M, N/64, 2]): SLIME_EPI_BIAS_RESID_F32_LN."""
    lib = _lib.load()
    M, K = a.shape
    N = (w if w is not None else w_frag).shape[0]
    x16 = torch.empty((M, N), dtype=a.dtype, device=a.device)
    stats = torch.empty((M, N // 64, 2), dtype=torch.float32, device=a.device)
    g = _lib.GemmArgs(A=_ptr(a), lda=a.stride(0), B=_ptr(w), bias=_ptr(bias), C=_ptr(h), ldc=h.stride(0), M=M, N=N, K=K,
                      dtype=dtype_code(a.dtype), epilogue=_lib.EPI_BIAS_RESID_F32_LN, x16=_ptr(x16), ldx=N, stats_out=_ptr(stats),
                      B_frag=_ptr(w_frag))
    _lib.check(lib.slime_gemm_ex(C.byref(g), _stream()), "slime_gemm_ex")
    return x16, stats


def _resid_shift(dtype: torch.dtype) -> int:
    if dtype == torch.bfloat16:
        return 8
    if dtype == torch.float16:
        return 5
    raise ValueError(f"the split residual stream's upper part is bf16 or fp16, not {dtype}")


def resid_split(h: torch.Tensor, dtype: torch.dtype):
    """fp32 rows -> the tower's SPLIT residual stream (include/slime_hip.h, ABI 7; a torch restatement of csrc/common.h resid_delta):
    hi = T(h) (RNE) and lo8 = floor((pattern(h) - pattern(float(hi))) / 2^SH) clamped to int8, SH = 8 (bf16) / 5 (fp16)."""
    sh = _resid_shift(dtype)
    hi = h.to(dtype)
    d = (h.float().contiguous().view(torch.int32) - hi.float().view(torch.int32)) >> sh          # int32 >> is arithmetic: floor
    return hi, d.clamp(-128, 127).to(torch.int8)


def resid_join(hi: torch.Tensor, lo8: torch.Tensor) -> torch.Tensor:
    """The fp32 rows a split residual stream holds (csrc/common.h resid_join): pattern(float(hi)) + lo8 2^SH + 2^(SH-1)."""
    sh = _resid_shift(hi.dtype)
    return (hi.float().contiguous().view(torch.int32) + lo8.to(torch.int32) * (1 << sh) + (1 << (sh - 1))).view(torch.float32)


def gemm_resid_split(a: torch.Tensor, w: Optional[torch.Tensor], bias: Optional[torch.Tensor], hi: torch.Tensor, lo: torch.Tensor,
                     w_frag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SLIME_EPI_BIAS_RESID_SPLIT_LN on the split residual stream (hi: T [M, N], lo: int8 [M, N], updated in place):
    c = a @ w.T + bias + resid_join(hi, lo); hi, lo = resid_split(c).  Returns stats [M, N/64, 2] (partial sums of c)."""
    assert lo.dtype == torch.int8 and lo.stride(1) == 1, "the stream's lower part is one signed byte per element (ABI 7)"
    lib = _lib.load()
    M, K = a.shape
    N = (w if w is not None else w_frag).shape[0]
    stats = torch.empty((M, N // 64, 2), dtype=torch.float32, device=a.device)
    g = _lib.GemmArgs(A=_ptr(a), lda=a.stride(0), B=_ptr(w), bias=_ptr(bias), C=_ptr(hi), ldc=hi.stride(0), M=M, N=N, K=K,
                      dtype=dtype_code(a.dtype), epilogue=_lib.EPI_BIAS_RESID_SPLIT_LN, stats_out=_ptr(stats), B_frag=_ptr(w_frag),
                      lo8=_ptr(lo), ldlo=lo.stride(0))
    _lib.check(lib.slime_gemm_ex(C.byref(g), _stream()), "slime_gemm_ex")
    return stats


def patch_embed_prenorm(pixels: torch.Tensor, patch_w_frag: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, ln_w: torch.Tensor,
                        ln_b: torch.Tensor, eps: float, dtype: torch.dtype, image: int, patch: int, kpad: int, want_h=True, want_x16=True,
                        want_lo=False):
    """slime_patch_embed_prenorm: pixels [n,3,image,image] (fp32 or T) -> (h fp32 [n*(1+P), D] | None, x16 T | None, lo int8 | None,
    stats [n*(1+P), D/64, 2] | None)."""
    lib = _lib.load()
    n, D = pixels.shape[0], cls.shape[0]
    rows = n * ((image // patch) ** 2 + 1)
    dev = pixels.device
    h = torch.empty((rows, D), dtype=torch.float32, device=dev) if want_h else None
    x16 = torch.empty((rows, D), dtype=dtype, device=dev) if want_x16 else None
    lo = torch.empty((rows, D), dtype=torch.int8, device=dev) if want_lo else None
    stats = torch.empty((rows, D // 64, 2), dtype=torch.float32, device=dev) if want_x16 else None
    _lib.check(lib.slime_patch_embed_prenorm(_ptr(pixels), dtype_code(pixels.dtype), _ptr(patch_w_frag), _ptr(cls), _ptr(pos), _ptr(ln_w),
                                             _ptr(ln_b), float(eps), _ptr(h), _ptr(x16), _ptr(lo), _ptr(stats), dtype_code(dtype), n, image,
                                             patch, kpad, D, _stream()), "slime_patch_embed_prenorm")
    return h, x16, lo, stats


def gemm_ln_consumer(x16: torch.Tensor, stats: torch.Tensor, w_folded: torch.Tensor, bias_folded: torch.Tensor, colsum: torch.Tensor,
                     eps: float, epilogue: int, w_frag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """epi(LayerNorm(x16) @ W.T + b) with the LayerNorm folded: w_folded = T(W diag(gamma)), bias_folded = b + W beta,
    colsum = row sums of w_folded; stats from the producer.  Output T [M, N]."""
    lib = _lib.load()
    M, K = x16.shape
    N = (w_folded if w_folded is not None else w_frag).shape[0]
    out = torch.empty((M, N), dtype=x16.dtype, device=x16.device)
    g = _lib.GemmArgs(A=_ptr(x16), lda=x16.stride(0), B=_ptr(w_folded), bias=_ptr(bias_folded), C=_ptr(out), ldc=N, M=M, N=N, K=K,
                      dtype=dtype_code(x16.dtype), epilogue=epilogue, ln_stats=_ptr(stats), ln_groups=stats.shape[1],
                      ln_colsum=_ptr(colsum), ln_eps=float(eps), B_frag=_ptr(w_frag))
    _lib.check(lib.slime_gemm_ex(C.byref(g), _stream()), "slime_gemm_ex")
    return out


def layernorm(x: torch.Tensor, w, b, eps: float, dtype: torch.dtype, want_f32=False, want_t=True,
              add: Optional[torch.Tensor] = None, normalize=True):
    lib = _lib.load()
    rows, D = x.shape
    o32 = torch.empty((rows, D), dtype=torch.float32, device=x.device) if want_f32 else None
    ot = torch.empty((rows, D), dtype=dtype, device=x.device) if want_t else None
    ot2 = torch.empty((rows, D), dtype=dtype, device=x.device) if add is not None else None
    _lib.check(lib.slime_layernorm(_ptr(x), x.stride(0), rows, D, _ptr(w), _ptr(b), float(eps), int(normalize),
                                   _ptr(o32), _ptr(ot), _ptr(ot2), _ptr(add), add.shape[0] if add is not None else 0,
                                   dtype_code(dtype), _stream()), "slime_layernorm")
    return o32, ot, ot2


def attention(q, k, v, heads: int, head_dim: int) -> torch.Tensor:
    """q [B|1, nq, H*dh] (pre-scaled by dh^-0.5 * log2 e), k/v [B, nkv, H*dh] -> [B, nq, H*dh]; all T, last dim contiguous."""
    lib = _lib.load()
    B, nkv = k.shape[0], k.shape[1]
    nq = q.shape[1]
    o = torch.empty((B, nq, heads * head_dim), dtype=k.dtype, device=k.device)
    for t in (q, k, v):
        assert t.stride(-1) == 1
    q_bs = 0 if q.shape[0] == 1 and B > 1 else q.stride(0)
    _lib.check(lib.slime_attention(q.data_ptr(), q_bs, q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                                   v.data_ptr(), v.stride(0), v.stride(1), o.data_ptr(), o.stride(0), o.stride(1),
                                   B, heads, head_dim, nq, nkv, dtype_code(k.dtype), _stream()), "slime_attention")
    return o


# ------------------------------------------------------------------------------------------------
# weight packing (load time; plain torch)
# ------------------------------------------------------------------------------------------------

@dataclass
class PackedTower:
    cfg: VisionConfig
    dtype: torch.dtype
    layers_run: int
    tensors: Dict[str, torch.Tensor]
    desc: _lib.VitDesc
    ws: Workspace = field(default_factory=Workspace)
    probe: Optional[tuple] = None        # (layer, kernel id, start event, stop event): in-situ kernel timing

    @property
    def device(self):
        return self.tensors["cls"].device


def layers_for_select(cfg: VisionConfig, select_layer: int) -> int:
    """hidden_states has L+1 entries; index i is produced by running i layers (clip_encoder.py:36-37)."""
    L = cfg.num_hidden_layers
    idx = select_layer if select_layer >= 0 else L + 1 + select_layer
    if not 0 <= idx <= L:
        raise ValueError(f"mm_vision_select_layer {select_layer} out of range for {L} layers")
    return idx


def pack_tower(state_dict: Dict[str, torch.Tensor], cfg: VisionConfig, dtype: torch.dtype, device,
               select_layer: int = -2) -> PackedTower:
    """HF CLIPVisionModel state dict (either key generation) -> device tensors in the kernels' layout."""
    if dtype not in (torch.bfloat16, torch.float16):
        raise ValueError("tower compute dtype must be bf16 or fp16 (MFMA operand type); fp32 I/O is supported")
    sd = strip_tower_prefix(state_dict)
    D, Fi, L = cfg.hidden_size, cfg.intermediate_size, layers_for_select(cfg, select_layer)
    P2 = 3 * cfg.patch_size * cfg.patch_size
    kpad = (P2 + 63) // 64 * 64
    scale = cfg.head_dim ** -0.5 * LOG2E                     # logits in log2 units (slime_attention contract)

    def f32(t):
        return t.detach().to(device=device, dtype=torch.float32).contiguous()

    def tt(t):
        return t.detach().to(device=device, dtype=torch.float32).to(dtype).contiguous()

    pw = torch.zeros((D, kpad), dtype=torch.float32)
    pw[:, :P2] = sd["embeddings.patch_embedding.weight"].detach().float().reshape(D, P2)
    T: Dict[str, torch.Tensor] = {
        "patch_w": tt(pw), "cls": f32(sd["embeddings.class_embedding"]),
        "pos": f32(sd["embeddings.position_embedding.weight"]),
        "pre_ln_w": f32(sd["pre_layrnorm.weight"]), "pre_ln_b": f32(sd["pre_layrnorm.bias"]),
    }

    def stack(fmt, conv, n=L):
        if n == 0:
            return None
        return torch.stack([conv(sd[fmt.format(i)]) for i in range(n)]).contiguous()

    if L > 0:
        p = "encoder.layers.{}."

        def fold_ln(Wm, bm, gamma, beta, row_scale=None):
            """LayerNorm folded into the Linear that consumes it (slime_gemm_ex): W' = W diag(gamma) rounded to T,
            b' = b + W beta, colsum = row sums of the ROUNDED W' (it must cancel the mean term of what the MFMA multiplies)."""
            W64 = Wm.detach().double().cpu()
            Wp = W64 * gamma.detach().double().cpu()[None, :]
            bp = bm.detach().double().cpu() + W64 @ beta.detach().double().cpu()
            if row_scale is not None:
                Wp, bp = Wp * row_scale[:, None], bp * row_scale
            Wt = Wp.float().to(dtype)
            return Wt, bp.float(), Wt.double().sum(1).float()

        rs = torch.ones(3 * D, dtype=torch.float64)
        rs[:D] = scale                                         # q rows carry dh^-0.5 * log2 e (slime_attention contract)
        wq, bq, cq, w1, b1, c1 = [], [], [], [], [], []
        for i in range(L):
            q = f"encoder.layers.{i}."
            a = q + "self_attn."
            Wm = torch.cat([sd[a + "q_proj.weight"].float(), sd[a + "k_proj.weight"].float(), sd[a + "v_proj.weight"].float()], 0)
            bm = torch.cat([sd[a + "q_proj.bias"].float(), sd[a + "k_proj.bias"].float(), sd[a + "v_proj.bias"].float()], 0)
            Wt, bp, cs = fold_ln(Wm, bm, sd[q + "layer_norm1.weight"], sd[q + "layer_norm1.bias"], rs)
            wq.append(Wt); bq.append(bp); cq.append(cs)
            Wt, bp, cs = fold_ln(sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"], sd[q + "layer_norm2.weight"], sd[q + "layer_norm2.bias"])
            w1.append(Wt); b1.append(bp); c1.append(cs)

        def dev_stack(ts):
            return torch.stack(ts).to(device).contiguous()

        T["w_qkv"], T["b_qkv"], T["colsum_qkv"] = dev_stack(wq), dev_stack(bq), dev_stack(cq)
        T["w_fc1"], T["b_fc1"], T["colsum_fc1"] = dev_stack(w1), dev_stack(b1), dev_stack(c1)
        T["w_o"], T["b_o"] = stack(p + "self_attn.out_proj.weight", tt), stack(p + "self_attn.out_proj.bias", f32)
        T["w_fc2"], T["b_fc2"] = stack(p + "mlp.fc2.weight", tt), stack(p + "mlp.fc2.bias", f32)
        # fragment-order images (the direct-B kernel loads them straight into registers, the LDS-staged kernels DMA from them);
        # the row-major tensors are dropped unless SLIME_KEEP_ROW_MAJOR=1
        pack_static(T, ("w_qkv", "w_o", "w_fc1", "w_fc2"))
    T["patch_w_frag"] = pack_b_frag(T["patch_w"])          # the front end (slime_patch_embed_prenorm) reads this image only
    if T["patch_w_frag"] is None:
        if not T["patch_w"].is_cuda:
            raise _lib.SlimeHipError("pack_tower: weights are packed on the GPU (slime_gemm_pack_b): slime_amd has no CPU path")
        raise ValueError("pack_tower: the patch-embed weight could not be packed (hidden size and padded patch size must be multiples of 64)")
    T["patch_w"] = None
    d = _lib.VitDesc()
    d.hidden, d.inter, d.heads, d.layers_run = D, Fi, cfg.num_attention_heads, L
    d.image, d.patch, d.kpad, d.dtype, d.eps = cfg.image_size, cfg.patch_size, kpad, dtype_code(dtype), cfg.layer_norm_eps
    for name in ("patch_w", "cls", "pos", "pre_ln_w", "pre_ln_b", "w_qkv", "b_qkv", "colsum_qkv", "w_o", "b_o",
                 "w_fc1", "b_fc1", "colsum_fc1", "w_fc2", "b_fc2", "w_qkv_frag", "w_o_frag", "w_fc1_frag", "w_fc2_frag", "patch_w_frag"):
        setattr(d, name, T[name].data_ptr() if name in T and T[name] is not None else None)
    # the driver's own validation, at PACK time: a geometry the fused front end cannot run (more than 24 patches per side, image
    # not a multiple of 8, ...) is refused here with the limit named, not by the first forward
    _lib.check(_lib.load().slime_vit_check(C.byref(d)), "pack_tower")
    return PackedTower(cfg, dtype, L, T, d)


def tower_forward(pt: PackedTower, pixels: torch.Tensor, out_dtype: Optional[torch.dtype] = None,
                  keep_cls: bool = False, want_hidden: bool = False, out: Optional[torch.Tensor] = None):
    """pixels [N,3,S,S] (fp32 or the tower dtype) -> features [N, P(+1), D] in out_dtype (written into ``out`` if given:
    a contiguous [N, P(+1), D] tensor or leading-dim slice of one)."""
    lib = _lib.load()
    _require_cuda(pixels, "pixels")
    cfg = pt.cfg
    if pixels.dim() != 4 or pixels.shape[1] != 3 or pixels.shape[2] != cfg.image_size or pixels.shape[3] != cfg.image_size:
        # HF raises on a size mismatch too (modeling_clip.py:204-207)
        raise ValueError(f"Input image size ({tuple(pixels.shape[2:])}) doesn't match model ({cfg.image_size}*{cfg.image_size}).")
    if pixels.dtype not in (torch.float32, pt.dtype):
        pixels = pixels.to(pt.dtype)
    pixels = pixels.contiguous()
    n = pixels.shape[0]
    out_dtype = out_dtype or pixels.dtype
    rows = cfg.seq_len if keep_cls else cfg.num_patches
    if out is None:
        out = torch.empty((n, rows, cfg.hidden_size), dtype=out_dtype, device=pixels.device)
    elif tuple(out.shape) != (n, rows, cfg.hidden_size) or out.dtype != out_dtype or not out.is_contiguous():
        raise ValueError("tower_forward: out must be a contiguous [N, rows, hidden] tensor of the output dtype")
    hidden = torch.empty((n, cfg.seq_len, cfg.hidden_size), dtype=torch.float32, device=pixels.device) if want_hidden else None
    need = lib.slime_vit_workspace_bytes(C.byref(pt.desc), n)
    ws = pt.ws.get(need, pixels.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    probe = None
    if pt.probe is not None:
        layer, kernel, e0, e1 = pt.probe
        probe = C.byref(_lib.Probe(int(layer), int(kernel), e0.cuda_event, e1.cuda_event))
    _lib.check(lib.slime_vit_forward_ex(C.byref(pt.desc), pixels.data_ptr(), dtype_code(pixels.dtype), n, out.data_ptr(),
                                        dtype_code(out_dtype), int(keep_cls), _ptr(hidden), base,
                                        ws.numel() - (base - ws.data_ptr()), _stream(), probe), "slime_vit_forward")
    return (out, hidden) if want_hidden else out


def tower_hidden_states(pt: PackedTower, pixels: torch.Tensor) -> torch.Tensor:
    """pixels [N,3,S,S] -> fp32 [layers_run + 1, N, 1 + P, D]: every hidden state of ONE tower pass (slime_vit_forward_states)."""
    lib = _lib.load()
    _require_cuda(pixels, "pixels")
    cfg = pt.cfg
    if pixels.dim() != 4 or pixels.shape[1] != 3 or pixels.shape[2] != cfg.image_size or pixels.shape[3] != cfg.image_size:
        raise ValueError(f"Input image size ({tuple(pixels.shape[2:])}) doesn't match model ({cfg.image_size}*{cfg.image_size}).")
    if pixels.dtype not in (torch.float32, pt.dtype):
        pixels = pixels.to(pt.dtype)
    pixels = pixels.contiguous()
    n = pixels.shape[0]
    states = torch.empty((pt.layers_run + 1, n, cfg.seq_len, cfg.hidden_size), dtype=torch.float32, device=pixels.device)
    need = lib.slime_vit_workspace_bytes(C.byref(pt.desc), n)
    ws = pt.ws.get(need, pixels.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    _lib.check(lib.slime_vit_forward_states(C.byref(pt.desc), pixels.data_ptr(), dtype_code(pixels.dtype), n, states.data_ptr(), base,
                                            ws.numel() - (base - ws.data_ptr()), _stream()), "slime_vit_forward_states")
    return states


def gemm_kernel_name(M: int, N: int, K: int, dtype: torch.dtype, epilogue: int, has_b_frag: bool = False) -> str:
    lib = _lib.load()
    buf = C.create_string_buffer(128)
    _lib.check(lib.slime_gemm_kernel_name(M, N, K, dtype_code(dtype), epilogue, int(has_b_frag), buf, 128), "slime_gemm_kernel_name")
    return buf.value.decode()


def tower_kernel_names(pt: PackedTower, n_crops: int) -> Dict[int, str]:
    """probe kernel id (include/slime_hip.h: slime_probe) -> rocprofv3 kernel name for a tower pass over n_crops crops."""
    cfg = pt.cfg
    M, D, Fi = n_crops * cfg.seq_len, cfg.hidden_size, cfg.intermediate_size
    t = "F16" if pt.dtype == torch.float16 else "BF16"
    fr = {k: pt.tensors.get(k + "_frag") is not None for k in ("w_qkv", "w_o", "w_fc1", "w_fc2")}
    resid_epi = _lib.load().slime_vit_residual_epilogue()      # what this build's tower launches for out_proj / fc2
    return {1: gemm_kernel_name(M, 3 * D, D, pt.dtype, _lib.EPI_BIAS_T, fr["w_qkv"]),
            2: f"attn64r_kernel<{t}, 3>" if 321 <= cfg.seq_len <= 608 and cfg.head_dim == 64 else f"attn_kernel<{t}, 64, 608, 8, 5>",
            5: gemm_kernel_name(M, Fi, D, pt.dtype, _lib.EPI_BIAS_QUICKGELU_T, fr["w_fc1"]),
            3: gemm_kernel_name(M, D, D, pt.dtype, resid_epi, fr["w_o"]),
            6: gemm_kernel_name(M, D, Fi, pt.dtype, resid_epi, fr["w_fc2"])}


@dataclass
class PackedResampler:
    dim: int
    heads: int
    n_query: int
    n_kv: int
    dtype: torch.dtype
    tensors: Dict[str, torch.Tensor]
    desc: _lib.ResamplerDesc
    ws: Workspace = field(default_factory=Workspace)


def _abs_pos(table: torch.Tensor, side: int) -> torch.Tensor:
    """get_abs_pos (sampler.py:27-36): bicubic resize of the square sincos table, rounded back to the
    table's dtype (fp16 in the reference)."""
    src = int(math.sqrt(table.shape[0]))
    dt = table.dtype
    return F.interpolate(table.float().reshape(1, src, src, -1).permute(0, 3, 1, 2), size=(side, side),
                         mode="bicubic", align_corners=False).permute(0, 2, 3, 1).flatten(0, 2).to(dt)


def pack_resampler(sd: Dict[str, torch.Tensor], dim: int, heads: int, n_kv: int, dtype: torch.dtype, device,
                   eps: float = 1e-6) -> PackedResampler:
    """Resampler state dict (keys as sampler.py:115-137) -> kernel layout.  The query side is input
    independent (sampler.py:161-163) and is folded into ``q_proj`` here, in fp32, once."""
    nq = sd["query"].shape[0]
    dh = dim // heads
    side = int(math.isqrt(n_kv))
    assert side * side == n_kv, "key grid must be square (sampler.py:146-147)"
    E = dim
    in_w, in_b = sd["attn.in_proj_weight"].float().cpu(), sd["attn.in_proj_bias"].float().cpu()
    pos_q = sd["pos_embed"].cpu()
    if torch.isnan(pos_q.float()).any():
        # sampler.py:150-154 ("some init error"): a stored table that contains NaN is regenerated from the sincos
        # formula in the stored dtype.  The reference tests this on the device at every call; here once, at pack time.
        from .weights import sincos_pos_embed_2d
        pos_q = torch.from_numpy(sincos_pos_embed_2d(dim, int(math.isqrt(nq)))).to(pos_q.dtype)
    q = F.layer_norm(sd["query"].float().cpu(), (E,), sd["ln_q.weight"].float().cpu(), sd["ln_q.bias"].float().cpu(), eps)
    q = F.linear(q + pos_q.float(), in_w[:E], in_b[:E]) * (dh ** -0.5 * LOG2E)
    pos_k = _abs_pos(pos_q, side).float()

    def f32(t):
        return t.detach().to(device=device, dtype=torch.float32).contiguous()

    def tt(t):
        return t.detach().to(device=device, dtype=torch.float32).to(dtype).contiguous()

    T = {"q_proj": tt(q), "pos_k": f32(pos_k), "ln_kv_w": f32(sd["ln_kv.weight"]), "ln_kv_b": f32(sd["ln_kv.bias"]),
         "w_k": tt(in_w[E:2 * E]), "b_k": f32(in_b[E:2 * E]), "w_v": tt(in_w[2 * E:]), "b_v": f32(in_b[2 * E:]),
         "w_o": tt(sd["attn.out_proj.weight"]), "b_o": f32(sd["attn.out_proj.bias"]),
         "ln_post_w": f32(sd["ln_post.weight"]), "ln_post_b": f32(sd["ln_post.bias"])}
    pack_static(T, ("w_k", "w_v", "w_o"))
    d = _lib.ResamplerDesc()
    d.dim, d.heads, d.n_query, d.n_kv, d.dtype, d.eps = dim, heads, nq, n_kv, dtype_code(dtype), eps
    for k, v in T.items():
        setattr(d, k, None if v is None else v.data_ptr())
    return PackedResampler(dim, heads, nq, n_kv, dtype, T, d)


def resampler_forward(pr: PackedResampler, x: torch.Tensor, want_t: bool = False):
    """x fp32 [n, n_kv, dim] -> fp32 [n, n_query, dim] (and the T copy if want_t)."""
    lib = _lib.load()
    _require_cuda(x, "x")
    x = x.float().contiguous()
    n = x.shape[0]
    assert x.shape[1] == pr.n_kv and x.shape[2] == pr.dim
    out = torch.empty((n, pr.n_query, pr.dim), dtype=torch.float32, device=x.device)
    out_t = torch.empty((n, pr.n_query, pr.dim), dtype=pr.dtype, device=x.device) if want_t else None
    need = lib.slime_resampler_workspace_bytes(C.byref(pr.desc), n)
    ws = pr.ws.get(need, x.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    _lib.check(lib.slime_resampler_forward(C.byref(pr.desc), x.data_ptr(), pr.dim, n, out.data_ptr(), _ptr(out_t), base,
                                           ws.numel() - (base - ws.data_ptr()), _stream()), "slime_resampler_forward")
    return (out, out_t) if want_t else out


@dataclass
class PackedMlp:
    in_dim: int
    hidden: int
    dtype: torch.dtype
    tensors: Dict[str, torch.Tensor]
    desc: _lib.MlpDesc
    ws: Workspace = field(default_factory=Workspace)


def pack_mlp(w1, b1, w2, b2, dtype: torch.dtype, device) -> PackedMlp:
    T = {"w1": w1.detach().to(device=device, dtype=torch.float32).to(dtype).contiguous(),
         "b1": b1.detach().to(device=device, dtype=torch.float32).contiguous(),
         "w2": w2.detach().to(device=device, dtype=torch.float32).to(dtype).contiguous(),
         "b2": b2.detach().to(device=device, dtype=torch.float32).contiguous()}
    pack_static(T, ("w1", "w2"))
    d = _lib.MlpDesc()
    d.in_dim, d.hidden, d.dtype = w1.shape[1], w1.shape[0], dtype_code(dtype)
    for k, v in T.items():
        setattr(d, k, None if v is None else v.data_ptr())
    return PackedMlp(w1.shape[1], w1.shape[0], dtype, T, d)


def mlp_forward(pm: PackedMlp, x: torch.Tensor) -> torch.Tensor:
    """x [rows, in_dim] fp32 or T -> fp32 [rows, hidden]."""
    lib = _lib.load()
    _require_cuda(x, "x")
    x = x.contiguous()
    rows = x.shape[0]
    out = torch.empty((rows, pm.hidden), dtype=torch.float32, device=x.device)
    need = lib.slime_mlp_workspace_bytes(C.byref(pm.desc), rows)
    ws = pm.ws.get(need, x.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    xf = x.data_ptr() if x.dtype == torch.float32 else None
    xt = x.data_ptr() if x.dtype == pm.dtype else None
    if xf is None and xt is None:
        x = x.float()
        xf = x.data_ptr()
    _lib.check(lib.slime_mlp_forward(C.byref(pm.desc), xf, xt, rows, out.data_ptr(), base,
                                     ws.numel() - (base - ws.data_ptr()), _stream()), "slime_mlp_forward")
    return out


@dataclass
class PackedGated:
    mlp: PackedMlp
    attn: PackedResampler
    w_gate: torch.Tensor
    ws: Workspace = field(default_factory=Workspace)


def pack_gated(sd: Dict[str, torch.Tensor], cfg: AdapterConfig, dtype: torch.dtype, device) -> PackedGated:
    """GatedBlock state dict (``mm_projector.`` prefix removed)."""
    mlp = pack_mlp(sd["projection.0.weight"], sd["projection.0.bias"], sd["projection.2.weight"],
                   sd["projection.2.bias"], dtype, device)
    attn = pack_resampler(sub_state(sd, "attn."), cfg.mm_hidden_size, cfg.num_heads, cfg.global_queries, dtype,
                          device, cfg.ln_eps)
    # w_gate is a bf16 Parameter cast to the activation dtype at use (projector/builder.py:148)
    wg = sd["w_gate"].detach().to(device=device, dtype=torch.float32).contiguous()
    return PackedGated(mlp, attn, wg)


def gated_forward(pg: PackedGated, x: torch.Tensor, learnable_gated: int = -1) -> torch.Tensor:
    """x fp32 [n, 576, D] -> fp32 [n, 576, H]: the full GatedBlock path."""
    lib = _lib.load()
    _require_cuda(x, "x")
    x = x.float().contiguous()
    n, Tn, D = x.shape
    assert Tn == pg.attn.n_kv and D == pg.mlp.in_dim
    out = torch.empty((n, Tn, pg.mlp.hidden), dtype=torch.float32, device=x.device)
    need = lib.slime_gated_workspace_bytes(C.byref(pg.mlp.desc), C.byref(pg.attn.desc), n)
    ws = pg.ws.get(need, x.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    _lib.check(lib.slime_gated_forward(C.byref(pg.mlp.desc), C.byref(pg.attn.desc), pg.w_gate.data_ptr(),
                                       int(learnable_gated), x.data_ptr(), n, out.data_ptr(), base,
                                       ws.numel() - (base - ws.data_ptr()), _stream()), "slime_gated_forward")
    return out


def adapter_forward(pg: PackedGated, post: Optional["PackedResampler"], feats: torch.Tensor, n_images: int, n_local: int,
                    nw: int, nh: int, merge: bool = True, learnable_gated: int = -1,
                    out_dtype: Optional[torch.dtype] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused adapter for images that share one crop layout: tower features T [n_images*(1+n_local), 576, D] ->
    tokens [n_images, 576 + n_local*g*g, H] (gated global rows, then the merged local rows).  ``out`` may be a
    wider [n_images, rows, H] buffer (extra rows per image are left untouched)."""
    lib = _lib.load()
    _require_cuda(feats, "feats")
    if feats.dtype != pg.mlp.dtype:
        feats = feats.to(pg.mlp.dtype)
    feats = feats.contiguous()
    assert feats.shape[0] == n_images * (1 + n_local) and feats.shape[1] == pg.attn.n_kv and feats.shape[2] == pg.mlp.in_dim
    rows = pg.attn.n_kv + (n_local * post.n_query if n_local else 0)
    if out is None:
        out = torch.empty((n_images, rows, pg.mlp.hidden), dtype=out_dtype or feats.dtype, device=feats.device)
    assert out.dim() == 3 and out.shape[0] == n_images and out.shape[1] >= rows and out.shape[2] == pg.mlp.hidden and out.is_contiguous()
    pdesc = C.byref(post.desc) if n_local else None
    need = lib.slime_adapter_workspace_bytes(C.byref(pg.mlp.desc), C.byref(pg.attn.desc), pdesc, n_images, n_local)
    ws = pg.ws.get(need, feats.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    _lib.check(lib.slime_adapter_forward(C.byref(pg.mlp.desc), C.byref(pg.attn.desc), pg.w_gate.data_ptr(), int(learnable_gated),
                                         pdesc, feats.data_ptr(), n_images, n_local, nw, nh, int(merge), out.data_ptr(),
                                         dtype_code(out.dtype), out.shape[1], base, ws.numel() - (base - ws.data_ptr()),
                                         _stream()), "slime_adapter_forward")
    return out


def adapter_forward_precompressed(pg: PackedGated, glob: torch.Tensor, comp: torch.Tensor, n_images: int, n_local: int, nw: int, nh: int,
                                  merge: bool = True, learnable_gated: int = -1, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Adapter for the compressed-local exchange (slime_amd.dist.sharded_tower_compressed): ``glob`` T [B, 576, D] tower
    features of the global views, ``comp`` T [B*n_local, q, D] local crops ALREADY through post_qformer.  Per-module sequence
    (GatedBlock, projection MLP, batched merge) -- the arithmetic of adapter_forward after its post_qformer stage."""
    lib = _lib.load()
    B, P, q = n_images, glob.shape[1], comp.shape[1]
    g = int(math.isqrt(q))
    H = pg.mlp.hidden
    out = torch.empty((B, P + n_local * q, H), dtype=out_dtype or glob.dtype, device=glob.device)
    gt = gated_forward(pg, glob.float(), learnable_gated)                                   # fp32 [B, P, H]
    loc = mlp_forward(pg.mlp, comp.reshape(-1, comp.shape[-1]).contiguous())               # fp32 [B*n_local*q, H]
    _lib.check(lib.slime_merge_rows_batched(gt.data_ptr(), P, out.data_ptr(), dtype_code(out.dtype), out.shape[1], 0, B, P, 1, 1, H, 0,
                                            _stream()), "slime_merge_rows_batched")
    _lib.check(lib.slime_merge_rows_batched(loc.data_ptr(), n_local * q, out.data_ptr(), dtype_code(out.dtype), out.shape[1], P, B, nw,
                                            nh, g, H, int(merge), _stream()), "slime_merge_rows_batched")
    return out


def merge_rows(local: torch.Tensor, out: torch.Tensor, dst_row0: int, nw: int, nh: int, grid: int, merge: bool):
    """Scatter [n, g*g, C] fp32 local tokens into ``out`` rows (spatial raster order or flat), casting."""
    lib = _lib.load()
    C_ = local.shape[-1]
    _lib.check(lib.slime_merge_rows(_ptr(local), _ptr(out), dtype_code(out.dtype), int(dst_row0), nw, nh, grid, C_,
                                    int(merge), _stream()), "slime_merge_rows")


def gather_rows(src: torch.Tensor, out: torch.Tensor, rows_in: int, row_off: int, groups: int, rows_out: int):
    lib = _lib.load()
    _lib.check(lib.slime_gather_rows(_ptr(src), rows_in, row_off, _ptr(out), dtype_code(out.dtype), groups, rows_out,
                                     src.shape[-1], _stream()), "slime_gather_rows")


def tile_normalize(canvas_u8: torch.Tensor, crop: int, mean, std, out_dtype: torch.dtype) -> torch.Tensor:
    """uint8 [Hc, Wc, 3] device canvas -> normalised crops [(Hc/crop)*(Wc/crop), 3, crop, crop]."""
    lib = _lib.load()
    _require_cuda(canvas_u8, "canvas")
    Hc, Wc, _ = canvas_u8.shape
    n = (Hc // crop) * (Wc // crop)
    out = torch.empty((n, 3, crop, crop), dtype=out_dtype, device=canvas_u8.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    _lib.check(lib.slime_tile_normalize(_ptr(canvas_u8.contiguous()), Hc, Wc, crop, m, s, out.data_ptr(),
                                        dtype_code(out_dtype), _stream()), "slime_tile_normalize")
    return out


def resample_tables(in_size: int, out_size: int):
    """Host-side Pillow coefficient tables (int32 numpy arrays): bounds [out, 2], kk [out, ksize]."""
    import numpy as np
    lib = _lib.load()
    ks = lib.slime_resample_ksize(in_size, out_size)
    bounds = np.empty((out_size, 2), dtype=np.int32)
    kk = np.empty((out_size, ks), dtype=np.int32)
    _lib.check(lib.slime_resample_coeffs(in_size, out_size, bounds.ctypes.data, kk.ctypes.data), "slime_resample_coeffs")
    return bounds, kk


_TABLE_CACHE = {}


def _device_tables(in_size: int, out_size: int, device: torch.device):
    key = (in_size, out_size, str(device))
    hit = _TABLE_CACHE.get(key)
    if hit is None:
        b, k = resample_tables(in_size, out_size)
        hit = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), k.shape[1])
        if len(_TABLE_CACHE) > 256:
            _TABLE_CACHE.clear()
        _TABLE_CACHE[key] = hit
    return hit


def resize_bicubic_u8(img_u8: torch.Tensor, out_w: int, out_h: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Bit-exact ``PIL.Image.resize((out_w, out_h))`` (default bicubic) of a uint8 [H, W, 3] device image.
    ``out`` may be a [out_h, out_w, 3] view into a larger canvas (last two dims contiguous)."""
    lib = _lib.load()
    _require_cuda(img_u8, "image")
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 3 or img_u8.shape[2] != 3:
        raise ValueError(f"resize_bicubic_u8 expects a uint8 [H, W, 3] image, got {img_u8.dtype} {tuple(img_u8.shape)}")
    if img_u8.stride(2) != 1 or img_u8.stride(1) != 3:
        img_u8 = img_u8.contiguous()
    H, W, _ = img_u8.shape
    if out is None:
        out = torch.empty((out_h, out_w, 3), dtype=torch.uint8, device=img_u8.device)
    elif tuple(out.shape) != (out_h, out_w, 3) or out.stride(2) != 1 or out.stride(1) != 3 or out.dtype != torch.uint8:
        raise ValueError("resize_bicubic_u8: out must be a uint8 [out_h, out_w, 3] view with packed pixels")
    bh = kh = bv = kv = None
    ksh = ksv = 0
    if W != out_w:
        bh, kh, ksh = _device_tables(W, out_w, img_u8.device)
    if H != out_h:
        bv, kv, ksv = _device_tables(H, out_h, img_u8.device)
    tmp = torch.empty((H * out_w * 3,), dtype=torch.uint8, device=img_u8.device) if (bh is not None and bv is not None) else None
    _lib.check(lib.slime_resize_bicubic_u8(img_u8.data_ptr(), H, W, img_u8.stride(0), out.data_ptr(), out.stride(0), out_h, out_w,
                                           _ptr(bh), _ptr(kh), ksh, _ptr(bv), _ptr(kv), ksv, _ptr(tmp),
                                           0 if tmp is None else tmp.numel(), _stream()), "slime_resize_bicubic_u8")
    return out


def resize_bicubic_u8_batched(imgs_u8: torch.Tensor, out_w: int, out_h: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Batched :func:`resize_bicubic_u8`: uint8 [B, H, W, 3] -> [B, out_h, out_w, 3], one launch per pass for the
    whole batch.  ``out`` may be the interior view canvas[:, y0:y0+out_h, x0:x0+out_w] of a [B, Hc, Wc, 3] buffer."""
    lib = _lib.load()
    _require_cuda(imgs_u8, "images")
    if imgs_u8.dtype != torch.uint8 or imgs_u8.dim() != 4 or imgs_u8.shape[3] != 3:
        raise ValueError(f"resize_bicubic_u8_batched expects uint8 [B, H, W, 3], got {imgs_u8.dtype} {tuple(imgs_u8.shape)}")
    imgs_u8 = imgs_u8.contiguous()
    B, H, W, _ = imgs_u8.shape
    if out is None:
        out = torch.empty((B, out_h, out_w, 3), dtype=torch.uint8, device=imgs_u8.device)
    elif (tuple(out.shape) != (B, out_h, out_w, 3) or out.stride(3) != 1 or out.stride(2) != 3 or out.dtype != torch.uint8):
        raise ValueError("resize_bicubic_u8_batched: out must be a uint8 [B, out_h, out_w, 3] view with packed pixels")
    bh = kh = bv = kv = None
    ksh = ksv = 0
    if W != out_w:
        bh, kh, ksh = _device_tables(W, out_w, imgs_u8.device)
    if H != out_h:
        bv, kv, ksv = _device_tables(H, out_h, imgs_u8.device)
    tmp = torch.empty((B * H * out_w * 3,), dtype=torch.uint8, device=imgs_u8.device) if (bh is not None and bv is not None) else None
    _lib.check(lib.slime_resize_bicubic_u8_batched(imgs_u8.data_ptr(), B, imgs_u8.stride(0), H, W, imgs_u8.stride(1), out.data_ptr(),
                                                   out.stride(0), out.stride(1), out_h, out_w, _ptr(bh), _ptr(kh), ksh, _ptr(bv),
                                                   _ptr(kv), ksv, _ptr(tmp), 0 if tmp is None else tmp.numel(), _stream()),
               "slime_resize_bicubic_u8_batched")
    return out


def tile_normalize_batched(canvas_u8: torch.Tensor, crop: int, mean, std, out: torch.Tensor, first_crop: int) -> None:
    """uint8 [B, Hc, Wc, 3] canvases -> normalised tiles written into out[b, first_crop + tile] of a
    [B, crops_per_image, 3, crop, crop] tensor (contiguous)."""
    lib = _lib.load()
    _require_cuda(canvas_u8, "canvas")
    canvas_u8 = canvas_u8.contiguous()
    B, Hc, Wc, _ = canvas_u8.shape
    tiles = (Hc // crop) * (Wc // crop)
    assert out.is_contiguous() and out.dim() == 5 and out.shape[0] == B and first_crop + tiles <= out.shape[1]
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    base = out.data_ptr() + first_crop * 3 * crop * crop * out.element_size()
    _lib.check(lib.slime_tile_normalize_batched(canvas_u8.data_ptr(), B, canvas_u8.stride(0), Hc, Wc, crop, m, s, base, out.shape[1],
                                                dtype_code(out.dtype), _stream()), "slime_tile_normalize_batched")


def router_scores(local_f: torch.Tensor, text: torch.Tensor, mask: Optional[torch.Tensor]) -> torch.Tensor:
    """Cosine router scores [T] (fp32) for local tokens [T,H] against text embeddings [L,H]."""
    lib = _lib.load()
    _require_cuda(local_f, "local_f")
    img = local_f.float().contiguous()
    txt = text.to(device=img.device, dtype=torch.float32).contiguous()
    T, H = img.shape
    L = txt.shape[0]
    m = None if mask is None else mask.to(device=img.device).ne(0).to(torch.uint8).contiguous()
    scores = torch.empty((T,), dtype=torch.float32, device=img.device)
    ws = torch.empty((L + H + 8,), dtype=torch.float32, device=img.device)
    _lib.check(lib.slime_router_scores(img.data_ptr(), T, txt.data_ptr(), L, _ptr(m), H, scores.data_ptr(), ws.data_ptr(),
                                       _stream()), "slime_router_scores")
    return scores


def router_select(scores: torch.Tensor, topp: float, temp: float, want_probs: bool = False):
    """Device-side top-p selection; returns (keep_idx_buffer [T] int32, count [1] int32[, probs])."""
    lib = _lib.load()
    T = scores.shape[0]
    keep = torch.empty((T,), dtype=torch.int32, device=scores.device)
    cnt = torch.empty((1,), dtype=torch.int32, device=scores.device)
    probs = torch.empty((T,), dtype=torch.float32, device=scores.device) if want_probs else None
    _lib.check(lib.slime_router_select(scores.data_ptr(), T, float(temp), float(topp), keep.data_ptr(), cnt.data_ptr(),
                                       _ptr(probs), _stream()), "slime_router_select")
    return (keep, cnt, probs) if want_probs else (keep, cnt)


def router_topp(local_f: torch.Tensor, text: torch.Tensor, mask: Optional[torch.Tensor], topp: float, temp: float) -> torch.Tensor:
    """Indices (ascending, int64) of the local tokens the text-guided router keeps.  One D2H read of the
    count -- the output length is data dependent (the reference syncs at ``nonzero`` too)."""
    if local_f.shape[0] == 0:
        return torch.zeros((0,), dtype=torch.long, device=local_f.device)
    keep, cnt = router_select(router_scores(local_f, text, mask), topp, temp)
    return keep[: int(cnt.item())].long()


def router_topp_batched(tokens: torch.Tensor, row_off: Sequence[int], n_rows: Sequence[int], text: torch.Tensor,
                        mask: Optional[torch.Tensor], topp: float, temp: float) -> List[torch.Tensor]:
    """The router for the B images of a step in ONE launch pair and ONE D2H read (of the B kept counts).
    ``tokens`` fp32 [rows, H] (any 2-D contiguous view of the token buffer); image b owns rows
    ``row_off[b] .. row_off[b] + n_rows[b] - 1``; ``text`` [B, L, H], ``mask`` [B, L] or None.
    Returns per image the kept LOCAL indices (ascending, int64, relative to row_off[b])."""
    lib = _lib.load()
    _require_cuda(tokens, "tokens")
    assert tokens.dim() == 2 and tokens.dtype == torch.float32 and tokens.is_contiguous()
    B, H = len(n_rows), tokens.shape[1]
    dev = tokens.device
    T_max = max(int(n) for n in n_rows) if B else 0
    if B == 0 or T_max == 0:
        return [torch.zeros((0,), dtype=torch.long, device=dev) for _ in range(B)]
    txt = text.to(device=dev, dtype=torch.float32).contiguous()
    L = txt.shape[1]
    m = None if mask is None else mask.to(device=dev).ne(0).to(torch.uint8).contiguous()
    off = torch.tensor([int(o) for o in row_off], dtype=torch.int64).to(dev, non_blocking=True)
    cnt_in = torch.tensor([int(n) for n in n_rows], dtype=torch.int32).to(dev, non_blocking=True)
    scores = torch.empty((B, T_max), dtype=torch.float32, device=dev)
    keep = torch.empty((B, T_max), dtype=torch.int32, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    ws = torch.empty((lib.slime_router_batched_workspace_floats(B, L, H) + 8,), dtype=torch.float32, device=dev)
    _lib.check(lib.slime_router_scores_batched(tokens.data_ptr(), off.data_ptr(), cnt_in.data_ptr(), B, T_max, txt.data_ptr(), L,
                                               _ptr(m), H, scores.data_ptr(), ws.data_ptr(), _stream()), "slime_router_scores_batched")
    _lib.check(lib.slime_router_select_batched(scores.data_ptr(), cnt_in.data_ptr(), B, T_max, float(temp), float(topp),
                                               keep.data_ptr(), cnt.data_ptr(), _stream()), "slime_router_select_batched")
    counts = cnt.cpu().tolist()                                   # the step's only host sync
    return [keep[b, :counts[b]].long() for b in range(B)]


# ------------------------------------------------------------------------------------------------
# after the visual tokens: splice + Llama prefill attention
# ------------------------------------------------------------------------------------------------

def splice_rows(table: Optional[torch.Tensor], feats: Optional[torch.Tensor], src: torch.Tensor, out_dtype: torch.dtype) -> torch.Tensor:
    """out[r] = table[src[r]] (src >= 0) | feats[-2 - src[r]] (src <= -2) | 0 (src == -1); src int64 [rows] on the device.
    One launch for the whole padded batch (llava_arch.py:343-459 builds it with per-sequence cat / split / stack)."""
    lib = _lib.load()
    _require_cuda(src, "src")
    H = (table if table is not None else feats).shape[-1]
    rows = src.numel()
    out = torch.empty((rows, H), dtype=out_dtype, device=src.device)
    t = None if table is None else table.contiguous()
    f = None if feats is None or feats.numel() == 0 else feats.contiguous()
    _lib.check(lib.slime_splice_rows(_ptr(t), dtype_code(t.dtype) if t is not None else 0, 0 if t is None else t.shape[0],
                                     _ptr(f), dtype_code(f.dtype) if f is not None else 0, 0 if f is None else f.shape[0],
                                     _ptr(src.contiguous()), out.data_ptr(), dtype_code(out_dtype), rows, H, _stream()),
               "slime_splice_rows")
    return out


@dataclass
class PackedLlamaAttention:
    hidden: int
    n_heads: int
    n_kv_heads: int
    head_dim: int
    dtype: torch.dtype
    tensors: Dict[str, torch.Tensor]
    desc: "_lib.LlamaAttnDesc"

    @property
    def ws(self) -> Workspace:
        """The layers of a language model run one after the other on one stream: they share ONE grow-only scratch buffer per
        (device, stream) (a Workspace per packed layer kept 32 x ~200 MB alive at the SliME-8B prefill shapes).  Keyed on the
        CURRENT stream: two models / requests prefilling on different streams (or devices) never write the same scratch, and a
        model split across GPUs keeps one buffer per device instead of re-allocating at every device switch."""
        return _llama_workspace()


# One grow-only Workspace per (device, stream) for the Llama attention sub-layer: ~200 MB at SliME-8B prefill shapes (8 x 1216 tokens).
# Bounded PER DEVICE: at most LLAMA_WS_MAX_STREAMS entries per device (SLIME_LLAMA_WS_MAX_STREAMS overrides the default of 4), the
# device's least recently used one evicted (a server that prefills on many torch streams would otherwise pin that much per stream
# handle forever -- the pool has ~32 handles per device; a model split across GPUs does not evict another device's buffers);
# ``release_llama_workspaces()`` drops all.  An evicted buffer goes back to the caching allocator.  It was allocated on the stream
# it served (the key's stream was current when Workspace.get grew it) and only ever used there, so the allocator's own
# allocation-stream ordering already keeps it from a new owner until that stream's queued work is done: no record_stream needed.
LLAMA_WS_MAX_STREAMS = max(1, int(os.environ.get("SLIME_LLAMA_WS_MAX_STREAMS", "4")))
_LLAMA_WS: "collections.OrderedDict[tuple, Workspace]" = collections.OrderedDict()
_LLAMA_WS_LOCK = threading.Lock()


def _llama_workspace() -> Workspace:
    st = torch.cuda.current_stream()
    key = (st.device.index, st.cuda_stream)
    with _LLAMA_WS_LOCK:
        ws = _LLAMA_WS.get(key)
        if ws is None:
            ws = _LLAMA_WS[key] = Workspace()
        _LLAMA_WS.move_to_end(key)
        mine = [k for k in _LLAMA_WS if k[0] == key[0]]            # this device's entries, least recently used first
        for k in mine[: max(0, len(mine) - LLAMA_WS_MAX_STREAMS)]:
            del _LLAMA_WS[k]
        return ws


def release_llama_workspaces() -> int:
    """Drop every cached Llama-attention workspace (returns the number of bytes released to the caching allocator)."""
    with _LLAMA_WS_LOCK:
        n = sum(w.buf.numel() for w in _LLAMA_WS.values() if w.buf is not None)
        _LLAMA_WS.clear()
    return n


def llama_inv_freq(head_dim: int, theta: float) -> torch.Tensor:
    """fp32 inverse frequencies exactly as LlamaRotaryEmbedding computes them (default rope type)."""
    return 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))


def pack_llama_attention(wq, wk, wv, wo, n_heads: int, n_kv_heads: int, dtype: torch.dtype, device,
                         rope_theta: float = 500000.0) -> PackedLlamaAttention:
    """q_proj / k_proj / v_proj / o_proj weights (nn.Linear layout, no biases) -> one fused [q|k|v] operand + o."""
    D = wq.shape[1]
    dh = wq.shape[0] // n_heads
    if dh != 128 or wk.shape[0] != n_kv_heads * dh or wv.shape[0] != n_kv_heads * dh or tuple(wo.shape) != (D, n_heads * dh):
        raise ValueError("pack_llama_attention: head_dim must be 128 and the projection shapes consistent")

    def tt(t):
        return t.detach().to(device=device, dtype=torch.float32).to(dtype).contiguous()

    T = {"w_qkv": tt(torch.cat([wq.detach().float().cpu(), wk.detach().float().cpu(), wv.detach().float().cpu()], 0)),
         "w_o": tt(wo), "inv_freq": llama_inv_freq(dh, rope_theta).to(device)}
    pack_static(T, ("w_qkv", "w_o"))
    d = _lib.LlamaAttnDesc()
    d.hidden, d.n_heads, d.n_kv_heads, d.head_dim, d.dtype = D, n_heads, n_kv_heads, dh, dtype_code(dtype)
    for k, v in T.items():
        setattr(d, k, None if v is None else v.data_ptr())
    return PackedLlamaAttention(D, n_heads, n_kv_heads, dh, dtype, T, d)


_RANGE_CACHE: Dict[str, object] = {}
_RANGE_LOCK = threading.Lock()


def token_ranges(attention_mask: Optional[torch.Tensor]):
    """Key-padding mask [B, S] -> (start, length) int32 [B] of the token run of every sequence.  Prefill masks are
    contiguous runs (right or left padding, llava_arch.py:435-455); anything else is rejected.  Host-side check: one D2H --
    per MASK TENSOR, not per layer: the 32 layers of a forward pass the same tensor object, so the last result is kept together
    with a reference to that object and reused while the caller hands in the very same object, unmodified (``is`` + the in-place
    version counter; a data pointer would not do -- the caching allocator recycles addresses)."""
    if attention_mask is None:
        return None, None
    with _RANGE_LOCK:                                    # one consistent (mask, version, start, length) record per look-up
        ref = _RANGE_CACHE.get("mask")
        if ref is not None and ref() is attention_mask and _RANGE_CACHE.get("version") == attention_mask._version:
            return _RANGE_CACHE["start"], _RANGE_CACHE["length"]
    start, length = _token_ranges(attention_mask)
    with _RANGE_LOCK:                                    # weak reference: the cache does not keep the last mask alive
        _RANGE_CACHE.update(mask=weakref.ref(attention_mask), version=attention_mask._version, start=start, length=length)
    return start, length


def _token_ranges(attention_mask: torch.Tensor):
    m = attention_mask.ne(0)
    length = m.sum(1)
    start = torch.where(length > 0, m.to(torch.int8).argmax(1), torch.zeros_like(length))
    S = m.shape[1]
    idx = torch.arange(S, device=m.device)[None]
    run = (idx >= start[:, None]) & (idx < (start + length)[:, None])
    if not bool(torch.equal(run, m)):
        raise ValueError("attention_mask must mark one contiguous run of tokens per sequence (left or right padding)")
    return start.to(torch.int32).contiguous(), length.to(torch.int32).contiguous()


def _position_ids(position_ids: Optional[torch.Tensor], B: int, S: int, device) -> torch.Tensor:
    """int32 [B, S] on the device.  HF passes position_ids of shape [1, S] (LlamaModel builds them from cache_position when
    the caller gives None, and prepare_inputs_labels_for_multimodal returns None in that case) and lets cos[position_ids]
    broadcast over the batch; the kernels index one entry per row, so the broadcast is materialised here."""
    if position_ids is None:
        position_ids = torch.arange(S, device=device)[None]
    if position_ids.dim() == 1:
        position_ids = position_ids[None]
    if position_ids.dim() != 2 or position_ids.shape[1] != S or position_ids.shape[0] not in (1, B):
        raise ValueError(f"position_ids of shape {tuple(position_ids.shape)} do not broadcast to ({B}, {S})")
    pos = position_ids.to(device=device, dtype=torch.int32).expand(B, S).contiguous()
    assert pos.shape == (B, S)
    return pos


def llama_attention_forward_resid(pa: PackedLlamaAttention, hidden: torch.Tensor, resid: torch.Tensor, position_ids: Optional[torch.Tensor] = None,
                                  attention_mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The attention sub-layer with the decoder layer's residual add in o_proj's epilogue (slime_llama_attn_forward_resid):
    out = T(resid + self_attn(hidden)), all T [B, S, D].  ``out`` may be ``resid`` itself (in-place stream); ``hidden`` may be
    ``resid`` (no norm in between) but not ``out``."""
    lib = _lib.load()
    _require_cuda(hidden, "hidden")
    B, S, D = hidden.shape
    assert D == pa.hidden and hidden.dtype == pa.dtype and resid.dtype == pa.dtype and tuple(resid.shape) == (B, S, D)
    x, r = hidden.contiguous(), resid.contiguous()
    pos = _position_ids(position_ids, B, S, x.device)
    start, length = token_ranges(None if attention_mask is None else attention_mask.to(x.device))
    if out is None:
        out = torch.empty((B, S, D), dtype=pa.dtype, device=x.device)
    assert out.is_contiguous() and out.dtype == pa.dtype and out.data_ptr() != x.data_ptr()
    need = lib.slime_llama_attn_workspace_bytes(C.byref(pa.desc), B, S)
    ws = pa.ws.get(need, x.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    _lib.check(lib.slime_llama_attn_forward_resid(C.byref(pa.desc), x.data_ptr(), pos.data_ptr(), _ptr(start), _ptr(length), B, S,
                                                  r.data_ptr(), out.data_ptr(), base, ws.numel() - (base - ws.data_ptr()), _stream()),
               "slime_llama_attn_forward_resid")
    return out


def llama_attention_forward(pa: PackedLlamaAttention, hidden: torch.Tensor, position_ids: Optional[torch.Tensor] = None,
                            attention_mask: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """hidden [B, S, D] -> [B, S, D]: q/k/v projection, RoPE, causal GQA attention over the un-padded tokens, o_proj."""
    lib = _lib.load()
    _require_cuda(hidden, "hidden")
    B, S, D = hidden.shape
    assert D == pa.hidden
    x = hidden.to(pa.dtype).contiguous()
    pos = _position_ids(position_ids, B, S, x.device)
    start, length = token_ranges(None if attention_mask is None else attention_mask.to(x.device))
    out_dtype = out_dtype or hidden.dtype
    kernel_out = torch.float32 if out_dtype == torch.float32 else pa.dtype
    out = torch.empty((B, S, D), dtype=kernel_out, device=x.device)
    need = lib.slime_llama_attn_workspace_bytes(C.byref(pa.desc), B, S)
    ws = pa.ws.get(need, x.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    _lib.check(lib.slime_llama_attn_forward(C.byref(pa.desc), x.data_ptr(), pos.data_ptr(), _ptr(start), _ptr(length), B, S,
                                            out.data_ptr(), dtype_code(kernel_out), base, ws.numel() - (base - ws.data_ptr()),
                                            _stream()), "slime_llama_attn_forward")
    return out if out.dtype == out_dtype else out.to(out_dtype)
