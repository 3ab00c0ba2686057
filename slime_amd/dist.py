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

from typing import Callable, List, Tuple

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


def sharded_tower(tower_fn: Callable[[torch.Tensor], torch.Tensor], crops: torch.Tensor, feat_shape: Tuple[int, int],
                  feat_dtype: torch.dtype = torch.float32, group=None) -> torch.Tensor:
    """Run ``tower_fn`` on this rank's block of ``crops`` ([N,3,S,S], identical on every rank) and
    all-gather the features: returns [N, *feat_shape] on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tower_fn(crops)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = crops.shape[0]
    lo, hi, per = shard_bounds(n, world, rank)
    local = torch.zeros((per,) + tuple(feat_shape), dtype=feat_dtype, device=crops.device)
    if hi > lo:
        local[: hi - lo] = tower_fn(crops[lo:hi].contiguous()).to(feat_dtype)
    gathered = torch.empty((world * per,) + tuple(feat_shape), dtype=feat_dtype, device=crops.device)
    dist.all_gather_into_tensor(gathered, local, group=group)
    return gathered[:n]


def sharded_tower_gather(local_feats: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """All-gather equally sized per-rank feature blocks [n,576,D] -> [world*n,576,D] (rank-major order)."""
    out = torch.empty((world * local_feats.shape[0],) + tuple(local_feats.shape[1:]), dtype=local_feats.dtype,
                      device=local_feats.device)
    dist.all_gather_into_tensor(out, local_feats.contiguous(), group=group)
    return out
