"""Crop-parallel sharding of the tower over the GPUs of one node.

The reference has no multi-GPU data path (its multi-GPU eval is N independent processes over
question chunks).  Every crop's ViT forward is independent, so the flat crop list of a batch is
block-partitioned over the ranks (one process per GPU), each rank runs the HIP tower on its block,
and ONE all-gather (``torch.distributed.all_gather_into_tensor``; backend ``nccl`` == RCCL over xGMI on
ROCm, ``gloo`` in the CPU tests) reassembles the [N, 576, D] feature tensor before the adapter.
Blocks are padded to ceil(N/G) crops with zero crops so the collective is a single fixed-size call;
the pad rows are dropped after the gather.  Because the tower is bit-wise batch-invariant
(tests/test_gpu_path.py::test_tower_batch_invariance) the gathered tensor is bit-identical to the
single-GPU result.
"""
from __future__ import annotations

import json
import os
import re
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int, int]:
    """(start, stop, per_rank): contiguous block of ceil(n/world) items per rank; trailing ranks may be
    short or empty."""
    per = -(-n_items // world)
    lo = min(rank * per, n_items)
    hi = min(lo + per, n_items)
    return lo, hi, per


def image_shard(n_images: int, world: int, rank: int) -> List[int]:
    """Images whose adapter work this rank owns (contiguous block)."""
    lo, hi, _ = shard_bounds(n_images, world, rank)
    return list(range(lo, hi))


# Tower latency curve the exchange policy decides from.  It is DATA, not code: a profile of one (device, model, dtype) --
# slime_amd/data/tower_latency_mi355x_vitl336_bf16.json by default, SLIME_TOWER_LATENCY_JSON for another file, or
# ``calibrate_tower_latency`` to measure the running configuration once.  On MI355X / ViT-L-336 / bf16: a pass costs ~2.4 ms however
# few crops it holds (23 layers x 5 dependent launches of 5-30 us kernels) and ~0.33 ms per crop beyond ~16 crops, where the GEMM
# grids fill the chip; from 8 crops on encode() runs two half batches on two streams.
_DEFAULT_PROFILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "tower_latency_mi355x_vitl336_bf16.json")
_profile_cache: Dict[str, dict] = {}


def tower_latency_profile(path: Optional[str] = None) -> dict:
    """The latency profile: {"device", "model", "dtype", "ms": {n: ms}, "ms_per_crop_beyond"}."""
    path = path or os.environ.get("SLIME_TOWER_LATENCY_JSON") or _DEFAULT_PROFILE
    if path not in _profile_cache:
        with open(path) as f:
            prof = json.load(f)
        prof["ms"] = {int(k): float(v) for k, v in prof["ms"].items()}
        _profile_cache[path] = prof
    return _profile_cache[path]


_DTYPE_KEYS = {"bf16": "bf16", "bfloat16": "bf16", "torch.bfloat16": "bf16", "fp16": "fp16", "f16": "fp16", "half": "fp16", "float16": "fp16",
               "torch.float16": "fp16", "fp32": "fp32", "f32": "fp32", "float": "fp32", "float32": "fp32", "torch.float32": "fp32"}


def _dtype_key(x) -> str:
    k = str(x).strip().lower()
    return _DTYPE_KEYS.get(k, k)


def _device_key(x) -> str:
    """'AMD Instinct MI355X' / 'MI355X' -> 'mi355x' (the part number is the key; an unrecognised name is compared whole).  Boxes whose
    marketing name is missing from the driver's id table report 'AMD Radeon Graphics': ``device_label`` appends the ISA name and
    the CU count, and 'gfx950' with 256 CUs is the MI355X."""
    t = str(x).lower()
    m = re.search(r"\bmi\d+[a-z]*\b", t)
    if m:
        return m.group(0)
    if "gfx950" in t and re.search(r"\b256 cus\b", t):
        return "mi355x"
    return t.strip()


def device_label(device=None) -> str:
    """Name of the running GPU for the profile applicability check: torch's device name plus ISA and CU count
    ('AMD Radeon Graphics gfx950 256 CUs' on boxes without the amdgpu.ids table)."""
    p = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device())
    arch = str(getattr(p, "gcnArchName", "")).split(":")[0]
    return f"{p.name} {arch} {p.multi_processor_count} CUs".strip()


def _model_key(x) -> str:
    """'CLIP-ViT-L/14-336 (23 live layers)' -> 'clip-vit-l/14-336': the first token names the tower, the rest is commentary."""
    parts = str(x).strip().lower().split()
    return parts[0] if parts else ""


def profile_applies(profile: dict, device_name: Optional[str] = None, model: Optional[str] = None, dtype: Optional[str] = None) -> bool:
    """Does the profile describe the configuration that is running?  Unknown (None) fields are not held against it; known ones
    are compared for EQUALITY on a canonical key (part number, tower name, bf16 / fp16 / fp32) -- a substring test would let
    'float16' pass for 'bfloat16' and 'MI35' for 'MI355X'."""
    def ok(want, have, key):
        return have is None or want is None or key(want) == key(have)
    return ok(profile.get("device"), device_name, _device_key) and ok(profile.get("model"), model, _model_key) \
        and ok(profile.get("dtype"), dtype, _dtype_key)


def calibrate_tower_latency(tower_fn: Callable[[torch.Tensor], torch.Tensor], make_crops: Callable[[int], torch.Tensor],
                            sizes: Sequence[int] = (1, 2, 3, 5, 7, 8, 9, 12, 17, 24, 34, 40), reps: int = 5) -> dict:
    """Measure the curve on the running device / model / dtype (one-off, a few hundred ms): returns a profile dict usable as
    ``choose_chunk(..., profile=...)`` or to be dumped as JSON for SLIME_TOWER_LATENCY_JSON."""
    ms = {}
    for n in sizes:
        x = make_crops(n)
        for _ in range(2):
            tower_fn(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            tower_fn(x)
        e1.record()
        torch.cuda.synchronize()
        ms[int(n)] = e0.elapsed_time(e1) / reps
    ks = sorted(ms)
    slope = (ms[ks[-1]] - ms[ks[-2]]) / (ks[-1] - ks[-2]) if len(ks) > 1 else 0.0
    return {"device": torch.cuda.get_device_name(), "model": None, "dtype": None, "ms": ms, "ms_per_crop_beyond": slope, "source": "calibrate_tower_latency"}


def tower_ms(n: int, profile: Optional[dict] = None) -> float:
    """Tower latency for n crops from the profile (piece-wise linear through its points, constant slope past the table)."""
    if n <= 0:
        return 0.0
    prof = profile or tower_latency_profile()
    tab = prof["ms"]
    ks = sorted(tab)
    if n >= ks[-1]:
        return tab[ks[-1]] + float(prof.get("ms_per_crop_beyond", 0.0)) * (n - ks[-1])
    for a, b in zip(ks, ks[1:]):
        if a <= n <= b:
            return tab[a] + (tab[b] - tab[a]) * (n - a) / (b - a)
    return tab[ks[0]]


def gather_ms(per_rank: int, world: int, bytes_per_crop: int = 576 * 1024 * 2, link_gb_s: float = 100.0) -> float:
    """All-gather time model for per_rank crops of bf16 features per rank: every rank receives (world - 1) blocks.  xGMI is
    point to point (7 links x ~153 GB/s per GPU): a ring is bound by ONE link, a direct exchange uses all of them; 100 GB/s of
    effective per-rank receive bandwidth is the conservative (ring) end, plus ~30 us of collective launch latency."""
    if world <= 1 or per_rank <= 0:
        return 0.0
    return 0.03 + (world - 1) * per_rank * bytes_per_crop / (link_gb_s * 1e6)


# fused adapter (GatedBlock + post_qformer + MLP + merge) for k images of 1 + 4 crops on one GPU, ms (profiles/r05_adapter_direct_store.txt
# and the per-rank rehearsals): the GatedBlock's 576-query Resampler and the stacked MLP are under-filled launches below 4 images
ADAPTER_MS = {0: 0.0, 1: 0.3, 2: 0.4, 4: 0.6, 8: 0.7}


def adapter_ms(images: int) -> float:
    ks = sorted(ADAPTER_MS)
    if images >= ks[-1]:
        return ADAPTER_MS[ks[-1]] * images / ks[-1]
    for a, b in zip(ks, ks[1:]):
        if a <= images <= b:
            return ADAPTER_MS[a] + (ADAPTER_MS[b] - ADAPTER_MS[a]) * (images - a) / (b - a)
    return 0.0


def predicted_step_ms(n_crops: int, crops_per_image: int, world: int, profile: Optional[dict] = None) -> float:
    """Modelled wall time of one STRONG-scaling step (DESIGN.md section 7's table, as a function): the tower over the rank's
    block of ceil(n_crops / world) crops (measured latency curve) + the all-gather of the tower features (transfer model above)
    + the adapter for the images the busiest rank owns.  Nothing is assumed to overlap: an upper bound on the step, hence a lower
    bound on the speed-up -- the figure bench.py prints beside the measured `strong` object, to be held against it."""
    per = -(-n_crops // world)
    images = n_crops // crops_per_image
    return tower_ms(per, profile) + gather_ms(per, world) + adapter_ms(-(-images // world))


def choose_chunk(per_rank: int, world: int, bytes_per_crop: int = 576 * 1024 * 2, link_gb_s: float = 100.0,
                 profile: Optional[dict] = None, device_name: Optional[str] = None, model: Optional[str] = None,
                 dtype: Optional[str] = None) -> int:
    """Micro-batch size for ``sharded_tower`` (0 = one tower pass + one all-gather).  Chunking hides all but the last
    micro-batch's transfer under the remaining tower work but pays the tower's per-pass floor once per extra micro-batch:
    it is chosen only where the modelled saving exceeds that cost -- and only when the latency profile describes the running
    configuration (``device_name`` / ``model`` / ``dtype``; otherwise 0).  With the MI355X curve that is never the case at the
    per-rank sizes of BASELINE configs 2 / 3 / 5 (5 ... 34 crops: a second pass costs 1.8-3 ms, the whole transfer <= 0.8 ms
    -- round 2's fixed chunk of 3 turned 9 crops from 4.8 into 8.2-9.7 ms); it starts to pay at ~100 crops per rank on 8 GPUs."""
    if world <= 1 or per_rank < 2:
        return 0
    profile = profile or tower_latency_profile()
    if not profile_applies(profile, device_name, model, dtype):
        return 0                                            # a curve of another device / tower / dtype decides nothing: one pass, one gather

    def tower_ms(n):                                        # noqa: F811 -- the curve of THIS profile
        return globals()["tower_ms"](n, profile)
    best, best_t = 0, tower_ms(per_rank) + gather_ms(per_rank, world, bytes_per_crop, link_gb_s)
    for k in (2, 3, 4):                                     # number of micro-batches
        c = -(-per_rank // k)
        sizes = [min(c, per_rank - i) for i in range(0, per_rank, c)]
        t = sum(tower_ms(x) for x in sizes) + gather_ms(sizes[-1], world, bytes_per_crop, link_gb_s)
        if t < best_t - 0.05:
            best, best_t = c, t
    return best


def sharded_tower(tower_fn: Callable[[torch.Tensor], torch.Tensor], crops: torch.Tensor, feat_shape: Tuple[int, int],
                  feat_dtype: torch.dtype = torch.float32, group=None, chunk: int = 0) -> torch.Tensor:
    """Run ``tower_fn`` on this rank's block of ``crops`` ([N,3,S,S], identical on every rank) and all-gather the
    features: returns [N, *feat_shape] on every rank.

    ``chunk`` > 0: the block is processed in micro-batches of ``chunk`` crops and every micro-batch is gathered with an
    asynchronous collective as soon as its features exist, so the xGMI transfer of micro-batch j runs under the tower of
    micro-batch j+1 (RCCL runs collectives on its own stream); only the last micro-batch's gather is exposed.  Same bytes,
    same result (the tower is batch invariant), one extra strided copy per micro-batch into the output tensor."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tower_fn(crops)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = crops.shape[0]
    lo, hi, per = shard_bounds(n, world, rank)
    dev = crops.device
    if chunk <= 0 or chunk >= per:
        local = torch.zeros((per,) + tuple(feat_shape), dtype=feat_dtype, device=dev)
        if hi > lo:
            local[: hi - lo] = tower_fn(crops[lo:hi].contiguous()).to(feat_dtype)
        gathered = torch.empty((world * per,) + tuple(feat_shape), dtype=feat_dtype, device=dev)
        dist.all_gather_into_tensor(gathered, local, group=group)
        return gathered[:n]
    out = torch.empty((world, per) + tuple(feat_shape), dtype=feat_dtype, device=dev)
    pending = []
    for j0 in range(0, per, chunk):                       # every rank walks the same micro-batches (short blocks pad with zeros)
        cj = min(chunk, per - j0)
        local = torch.zeros((cj,) + tuple(feat_shape), dtype=feat_dtype, device=dev)
        a, b = lo + j0, min(lo + j0 + cj, hi)
        if b > a:
            local[: b - a] = tower_fn(crops[a:b].contiguous()).to(feat_dtype)
        buf = torch.empty((world * cj,) + tuple(feat_shape), dtype=feat_dtype, device=dev)
        work = dist.all_gather_into_tensor(buf, local, group=group, async_op=True)
        pending.append((work, buf, j0, cj, local))
    for work, buf, j0, cj, _keep in pending:
        work.wait()
        out[:, j0:j0 + cj] = buf.view(world, cj, *feat_shape)
    return out.view(world * per, *feat_shape)[:n]


def sharded_tower_compressed(tower_fn: Callable[[torch.Tensor], torch.Tensor], compress_fn: Callable[[torch.Tensor], torch.Tensor],
                             crops: torch.Tensor, crops_per_image: int, feat_shape: Tuple[int, int], n_query: int,
                             feat_dtype: torch.dtype = torch.bfloat16, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """SURVEY.md section 8e alternative: ``post_qformer`` is per-crop independent too, so every rank compresses the LOCAL crops
    of its block before the exchange and the collective moves [., n_query, D] (144 rows) per local crop instead of the full
    [., 576, D]; global crops (crop 0 of every ``crops_per_image``) travel uncompressed for the GatedBlock.  4x fewer bytes
    for the local crops: (1 + 16) crops -> 576 + 16*144 rows instead of 17*576 (0.29x); (1 + 4) -> 0.40x.
    Returns (global feats [n_images, *feat_shape], compressed local feats [n_local, n_query, D]) in crop order on every rank."""
    n = crops.shape[0]
    P, Dm = feat_shape
    dev = crops.device
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        feats = tower_fn(crops).to(feat_dtype)
        is_g = torch.arange(n, device=dev) % crops_per_image == 0
        return feats[is_g], compress_fn(feats[~is_g]).to(feat_dtype)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = []
    for r in range(world):
        lo_r, hi_r, _ = shard_bounds(n, world, r)
        g = sum(1 for c in range(lo_r, hi_r) if c % crops_per_image == 0)
        counts.append((g, (hi_r - lo_r) - g))
    gmax, lmax = max(c[0] for c in counts), max(c[1] for c in counts)
    lo, hi, _ = shard_bounds(n, world, rank)
    g_loc = torch.zeros((max(gmax, 1), P, Dm), dtype=feat_dtype, device=dev)
    l_loc = torch.zeros((max(lmax, 1), n_query, Dm), dtype=feat_dtype, device=dev)
    if hi > lo:
        feats = tower_fn(crops[lo:hi].contiguous()).to(feat_dtype)
        is_g = (torch.arange(lo, hi, device=dev) % crops_per_image) == 0
        if counts[rank][0]:
            g_loc[: counts[rank][0]] = feats[is_g]
        if counts[rank][1]:
            l_loc[: counts[rank][1]] = compress_fn(feats[~is_g]).to(feat_dtype)
    g_all = torch.empty((world * g_loc.shape[0], P, Dm), dtype=feat_dtype, device=dev)
    l_all = torch.empty((world * l_loc.shape[0], n_query, Dm), dtype=feat_dtype, device=dev)
    wg = dist.all_gather_into_tensor(g_all, g_loc, group=group, async_op=True)
    wl = dist.all_gather_into_tensor(l_all, l_loc, group=group, async_op=True)
    wg.wait(); wl.wait()
    g_all = g_all.view(world, g_loc.shape[0], P, Dm)
    l_all = l_all.view(world, l_loc.shape[0], n_query, Dm)
    glob = torch.cat([g_all[r, : counts[r][0]] for r in range(world)], 0)
    comp = torch.cat([l_all[r, : counts[r][1]] for r in range(world)], 0)
    return glob, comp


def gather_bytes(n_crops: int, crops_per_image: int, world: int, P: int = 576, n_query: int = 144, D: int = 1024, elem: int = 2):
    """(bytes received per rank: full-feature gather, compressed-local gather) for the DESIGN.md section 7 table."""
    per = -(-n_crops // world)
    full = (world - 1) * per * P * D * elem
    n_img = n_crops // crops_per_image
    gper, lper = -(-n_img // world), -(-(n_crops - n_img) // world)
    return full, (world - 1) * (gper * P + lper * n_query) * D * elem


def sharded_tower_gather(local_feats: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """All-gather equally sized per-rank feature blocks [n,576,D] -> [world*n,576,D] (rank-major order)."""
    out = torch.empty((world * local_feats.shape[0],) + tuple(local_feats.shape[1:]), dtype=local_feats.dtype,
                      device=local_feats.device)
    dist.all_gather_into_tensor(out, local_feats.contiguous(), group=group)
    return out


# ------------------------------------------------------------------------------------------------------------------------
# Head-sharded prefill attention (beyond the crop split north_star names -> opt-in: bench.py --prefill-shard heads)
# ------------------------------------------------------------------------------------------------------------------------
# Configs 4 / 5 replicate the 32 attention sub-layers on every rank (Amdahl: ~44 of 60 ms do not shrink with more GPUs).  The
# natural shard of grouped-query attention is the kv head: rank r owns n_kv_heads / G kv heads and their query heads --
#   q/k/v GEMM: column slice of the fused [q|k|v] operand (N = 6144 / G at Llama-3-8B), no communication;
#   RoPE + causal GQA attention: local to the rank's heads;
#   o_proj: ROW-parallel -- rank r multiplies its ctx slice [M, HQ/G * 128] with the matching 512 columns of W_o and holds a
#           partial [M, D]; ONE all-reduce (sum) per layer completes it.  The decoder layer's residual is folded into rank 0's
#           partial (its o_proj epilogue adds it), so the reduced tensor is the next layer's input: no arithmetic outside the
#           library, one collective per layer.
# Bytes: the all-reduce moves M * D * 2 B of 16-bit partials per layer (config 5: 9280 x 4096 x 2 = 76 MB; config 4: 80 MB);
# with RCCL's direct (all-links) algorithm on a fully connected xGMI node every rank sends and receives 2 * (G-1)/G of that over
# 7 links: ~0.13 ms per layer at 8 GPUs against ~0.22 ms of sharded compute (1.73 ms / 8) -- the exchange is NOT hideable (the
# next layer's q/k/v GEMM consumes the reduced rows), so the prefill part scales ~4-5x on 8 GPUs instead of 1x (replicated).
# Numerics: the partial sums are rounded to T before the reduction (standard tensor parallelism): tolerance-equal, not bit-equal,
# to the replicated result -- which is why the replicated form stays the default until a multi-GPU box has measured both.

def shard_llama_attention_weights(wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor, wo: torch.Tensor, n_heads: int,
                                  n_kv_heads: int, world: int, rank: int, head_dim: int = 128):
    """Rank ``rank``'s slice of one attention layer: (wq_r, wk_r, wv_r, wo_r, n_heads_r, n_kv_heads_r).  kv heads are the unit:
    ``n_kv_heads % world == 0`` (Llama-3-8B: 8 kv heads -> 1 per GPU on a node)."""
    if n_kv_heads % world != 0 or n_heads % n_kv_heads != 0:
        raise ValueError(f"{n_kv_heads} kv heads do not split over {world} ranks")
    kv = n_kv_heads // world
    grp = n_heads // n_kv_heads
    qs = slice(rank * kv * grp * head_dim, (rank + 1) * kv * grp * head_dim)
    ks = slice(rank * kv * head_dim, (rank + 1) * kv * head_dim)
    return wq[qs], wk[ks], wv[ks], wo[:, qs], kv * grp, kv


def head_sharded_attention(local_attn: Callable[[torch.Tensor, "torch.Tensor | None"], torch.Tensor], hidden: torch.Tensor,
                           resid: torch.Tensor, group=None) -> torch.Tensor:
    """One decoder-layer attention step over head-sharded weights: ``local_attn(hidden, resid_or_None)`` returns this rank's
    partial ``[B, S, D]`` (o_proj over the rank's heads; rank 0 is handed ``resid`` and folds it in, the other ranks get None),
    the all-reduce sums the partials in place.  Returns ``resid + self_attn(hidden)`` on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_attn(hidden, resid)
    part = local_attn(hidden, resid if dist.get_rank(group) == 0 else None).contiguous()
    dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return part


def head_shard_bytes(rows: int, hidden: int = 4096, elem: int = 2) -> int:
    """Bytes of one layer's all-reduce operand (the 16-bit partial output rows) for the DESIGN.md section 7 table."""
    return rows * hidden * elem
