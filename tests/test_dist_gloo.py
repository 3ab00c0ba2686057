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
        from slime_amd.dist import sharded_tower, shard_bounds, sharded_tower_gather
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
        eq = sharded_tower_gather(torch.full((3, 2, 2), float(rank)), world)
        ok = ok and eq.shape[0] == 3 * world and all(float(eq[3 * r].mean()) == r for r in range(world))
        q.put((rank, bool(ok), lo, hi, per))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_crops", [5, 4, 1])
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
