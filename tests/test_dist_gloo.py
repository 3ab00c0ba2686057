"""World-size-2 gloo tests of the crop sharding + all-gather logic (CPU, no GPU).  The tower is replaced
by the CPU oracle here -- as the CHECKER of the sharding code, never as a product path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_crops, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from slime_amd import weights as W
        from slime_amd.dist import sharded_tower, shard_bounds, sharded_tower_gather, sharded_tower_compressed
        from oracle import slime_oracle as O
        torch.set_num_threads(2)
        cfg = W.VisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=1,
                             image_size=56, patch_size=14)
        sd = W.strip_tower_prefix(W.make_tower_state_dict(cfg, seed=5))
        px = W.synthetic_pixels(n_crops, seed=9, image_size=56)
        calls = []

        def tower(x):
            calls.append(x.shape[0])
            return O.tower_forward(sd, cfg, x)

        full = O.tower_forward(sd, cfg, px)
        got = sharded_tower(tower, px, (16, 64))
        lo, hi, per = shard_bounds(n_crops, world, rank)
        ok = torch.equal(got, full) and calls == ([hi - lo] if hi > lo else [])
        # chunked (asynchronous, overlappable) gather: same tensor
        for chunk in (1, 2):
            ok = ok and torch.equal(sharded_tower(lambda x: O.tower_forward(sd, cfg, x), px, (16, 64), chunk=chunk), full)
        # compressed-local variant: globals [., 16, 64] + pooled locals [., 4, 64] in crop order (a toy per-crop compress_fn)
        per_image = 3 if n_crops % 3 == 0 else n_crops

        def compress(x):
            return x.view(x.shape[0], 4, 4, 64).mean(2)

        glob, comp = sharded_tower_compressed(lambda x: O.tower_forward(sd, cfg, x), compress, px, per_image, (16, 64), 4,
                                              torch.float32)
        is_g = torch.arange(n_crops) % per_image == 0
        ok = ok and torch.equal(glob, full[is_g]) and torch.equal(comp, compress(full[~is_g]))
        eq = sharded_tower_gather(torch.full((3, 2, 2), float(rank)), world)
        ok = ok and eq.shape[0] == 3 * world and all(float(eq[3 * r].mean()) == r for r in range(world))
        q.put((rank, bool(ok), lo, hi, per))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_crops", [5, 4, 1, 6])
def test_sharded_tower_world2(n_crops):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_crops, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, *_ in res), res
    covered = sum(hi - lo for _, _, lo, hi, _ in res)
    assert covered == n_crops


def test_shard_bounds_cover_everything():
    from slime_amd.dist import shard_bounds, image_shard
    for n in (1, 5, 40, 68, 320):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                lo, hi, per = shard_bounds(n, world, r)
                assert 0 <= hi - lo <= per
                seen.extend(range(lo, hi))
            assert seen == list(range(n))
    assert image_shard(8, 8, 3) == [3] and image_shard(4, 8, 7) == []


def test_gather_bytes_table():
    """bytes received per rank, full-feature gather vs compressed-local gather."""
    from slime_amd.dist import gather_bytes
    full, comp = gather_bytes(68, 17, 8)
    assert full == 7 * 9 * 576 * 1024 * 2 and comp == 7 * (1 * 576 + 8 * 144) * 1024 * 2
    assert 0.25 < comp / full < 0.35
    full, comp = gather_bytes(320, 5, 8)
    assert comp / full < 0.45


def _head_shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from slime_amd.dist import shard_llama_attention_weights, head_sharded_attention
        from oracle import prefill_oracle as P
        torch.set_num_threads(2)
        D, HQ, HKV, B, S = 1024, 8, 4, 2, 37
        g = torch.Generator().manual_seed(17)
        w = [torch.randn(n, k, generator=g) * k ** -0.5 for n, k in ((HQ * 128, D), (HKV * 128, D), (HKV * 128, D), (D, HQ * 128))]
        hidden = torch.randn(B, S, D, generator=g)
        resid = torch.randn(B, S, D, generator=g)
        mask = torch.ones(B, S, dtype=torch.int64)
        mask[1, 30:] = 0
        pos = torch.arange(S)[None].expand(B, S)
        full = resid + P.llama_attention_forward(hidden, w[0], w[1], w[2], w[3], HQ, HKV, pos, mask)
        wq, wk, wv, wo, hq_r, hkv_r = shard_llama_attention_weights(w[0], w[1], w[2], w[3], HQ, HKV, world, rank)
        calls = []

        def local(h, r):
            calls.append(r is not None)
            part = P.llama_attention_forward(h, wq, wk, wv, wo, hq_r, hkv_r, pos, mask, head_dim=128)       # the oracle as the rank-local kernel
            return part + r if r is not None else part

        got = head_sharded_attention(local, hidden, resid)
        err = float((got - full).norm() / full.norm())
        ok = err < 1e-5 and calls == [rank == 0] and hq_r == HQ // world and hkv_r == HKV // world
        ok = ok and tuple(wq.shape) == (HQ // world * 128, D) and tuple(wo.shape) == (D, HQ // world * 128)
        q.put((rank, bool(ok), err))
    finally:
        dist.destroy_process_group()


def test_head_sharded_prefill_attention_world2():
    """kv-head sharding of the Llama attention sub-layer (column-sliced q/k/v, local causal GQA, row-parallel
    o_proj + one all-reduce, residual folded into rank 0's partial) equals the replicated result to fp32 rounding.  The oracle
    stands in for the rank-local HIP call (checker of the sharding code, not a product path)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_head_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def test_head_shard_weight_slices_cover_the_layer():
    from slime_amd.dist import shard_llama_attention_weights, head_shard_bytes
    HQ, HKV, D = 32, 8, 256
    wq, wk, wv, wo = torch.arange(HQ * 128 * D).view(HQ * 128, D), torch.arange(HKV * 128 * D).view(HKV * 128, D), \
        -torch.arange(HKV * 128 * D).view(HKV * 128, D), torch.arange(D * HQ * 128).view(D, HQ * 128)
    for world in (1, 2, 4, 8):
        parts = [shard_llama_attention_weights(wq, wk, wv, wo, HQ, HKV, world, r) for r in range(world)]
        assert torch.equal(torch.cat([p[0] for p in parts]), wq) and torch.equal(torch.cat([p[1] for p in parts]), wk)
        assert torch.equal(torch.cat([p[2] for p in parts]), wv) and torch.equal(torch.cat([p[3] for p in parts], 1), wo)
        assert all(p[4] == HQ // world and p[5] == HKV // world for p in parts)
    with pytest.raises(ValueError):
        shard_llama_attention_weights(wq, wk, wv, wo, HQ, HKV, 3, 0)
    assert head_shard_bytes(9280) == 9280 * 4096 * 2


def test_predicted_step_model_matches_design_table():
    """slime_amd.dist.predicted_step_ms (what bench.py prints beside the measured `strong` object): the committed MI355X latency
    curve + transfer model + adapter share reproduce DESIGN.md section 7's strong-scaling rows and are monotone in the rank count."""
    from slime_amd import dist as D
    t = {n: D.predicted_step_ms(40, 5, n) for n in (1, 2, 4, 8)}
    assert t[1] > t[2] > t[4] > t[8] > 0
    assert abs(t[1] - (D.tower_ms(40) + D.adapter_ms(8))) < 1e-9                # one GPU: no gather
    assert abs(t[8] - (D.tower_ms(5) + D.gather_ms(5, 8) + D.adapter_ms(1))) < 1e-9
    assert 3.5 < t[1] / t[8] < 5.0                                               # 5 crops per rank run far below the 40-crop rate
    c3 = {n: D.predicted_step_ms(68, 17, n) for n in (1, 8)}
    assert 4.0 < c3[1] / c3[8] < 5.5
    assert D.adapter_ms(0) == 0.0 and D.adapter_ms(3) == (D.ADAPTER_MS[2] + D.ADAPTER_MS[4]) / 2 and D.adapter_ms(16) == 2 * D.ADAPTER_MS[8]
